#!/bin/bash
# round 4, second measurement set: the sources with the lane encoder's chunked, measured table slab (the decoder's kernels are unchanged, so its
# SQ-counter and D3-traffic files of the first set stay): GPU tests, default bench line, rocprofv3 kernel stats, PMC traffic of the three operations
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_final2; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -25 > $O/gpu_tests.txt
tail -3 $O/gpu_tests.txt
timeout 900 python bench.py > $O/bench_default_full.json 2> $O/bench_default_full.err
tail -c 400 $O/bench_default_full.json; echo
bash tools/profile_bench.sh r04_final2/prof > $O/profile.log 2>&1
cat $O/prof/kernel_stats_summary.txt | head -12
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic/pmc_traffic.json $O/ 2>/dev/null; tail -2 $O/pmc_traffic.log
