#!/bin/bash
# round 6, call 1: the hygiene commit (LDS-store probe + fallback, build id, bench changes) on the GPU: every GPU test, smoke, the driver's bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call1; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 > $O/gpu_tests_full_suite.txt
tail -3 $O/gpu_tests_full_suite.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
tail -5 $O/bench_driver_style.err
python -c "import json;d=json.load(open('$O/bench_driver_style.json'));print('driver style:',d['value'],d['roofline']['frac'],d['roofline_encode']['uncompressed_GBps'],d['roofline_hc']['uncompressed_GBps'],d['library_build_id'],d['cpu_baseline'])"
