#!/bin/bash
# what the driver runs at round end, on the final tree: GPU tests, smoke, the bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_final_sanity; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3 | tee $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench_driver_style.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_final_sanity/bench_driver_style.json'))
print('value',d['value'],'frac',d['roofline']['frac'],'traffic',d['roofline'].get('traffic'),'enc',d.get('roofline_encode',{}).get('achieved_uncompressed_GBps', d.get('roofline_encode',{})), )
PY
