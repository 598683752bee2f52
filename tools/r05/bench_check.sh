#!/bin/bash
# bench.py after adding extras.batch_size_sweep + the new GPU test (no kernel change)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_bench_check; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "lds_out_of_range" 2>&1 | tail -8 | tee $O/lds_test.txt
timeout 1200 python bench.py > $O/bench_default_full.json 2> $O/bench.err; tail -c 200 $O/bench_default_full.json; echo
python -c "
import json;d=json.load(open('$O/bench_default_full.json'))
print(d['value'],d['roofline']['frac'],d['roofline']['traffic'])
for k,v in d['extras'].items():
    if 'sweep' in k or 'host' in k: print(k,v)"
