#!/bin/bash
# round 6, THE measurement set (after the last kernel change): every GPU test, smoke(), the default bench line, the driver's bench command,
# rocprofv3 kernel stats of the headline workload, PMC traffic of the three timed operations, SQ counters of the wavefront kernels on 512 blocks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_final; rm -rf $O; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -30 > $O/gpu_tests_full_suite.txt
tail -3 $O/gpu_tests_full_suite.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python bench.py > $O/bench_default_full.json 2> $O/bench_default_full.err
tail -c 300 $O/bench_default_full.json; echo
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
python -c "import json;d=json.load(open('$O/bench_driver_style.json'));print('driver style:',d['value'],d['roofline']['frac'],d['roofline_encode']['uncompressed_GBps'],d['roofline_hc']['uncompressed_GBps'],d['library_build_id'])"
bash tools/profile_bench.sh r06_final/prof > $O/profile.log 2>&1
head -8 $O/prof/kernel_stats_summary.txt
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic/pmc_traffic.json $O/ 2>/dev/null; tail -2 $O/pmc_traffic.log
bash tools/pmc_wave_small_batch.sh 512 > $O/pmc_wave_small_batch.log 2>&1
cp gpurun_out/pmc_wave_small_batch/summary.json $O/pmc_wave_kernels_512_blocks.json 2>/dev/null; tail -5 $O/pmc_wave_small_batch.log
