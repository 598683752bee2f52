#!/bin/bash
# round 4: the measurement set of the final sources (one call): GPU tests, default bench line, rocprofv3 kernel stats, PMC traffic, SQ counters
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_final; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -25 > $O/gpu_tests.txt
tail -3 $O/gpu_tests.txt
timeout 900 python bench.py > $O/bench_default_full.json 2> $O/bench_default_full.err
tail -c 600 $O/bench_default_full.json; echo
bash tools/profile_bench.sh r04_final/prof > $O/profile.log 2>&1
cat $O/prof/kernel_stats_summary.txt | head -12
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic/pmc_traffic.json $O/ 2>/dev/null; tail -2 $O/pmc_traffic.log
PMC_QUICK=1 bash tools/pmc_decoder.sh r04_final_sq 2 1048576 > $O/pmc_sq.log 2>&1; cp gpurun_out/pmc_r04_final_sq/summary.json $O/pmc_decode_lane4_D2_sq.json; tail -1 $O/pmc_sq.log
bash tools/pmc_decode_traffic.sh 3 > $O/pmc_decode_traffic_D3.log 2>&1; tail -2 $O/pmc_decode_traffic_D3.log
