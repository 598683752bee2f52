#!/bin/bash
# round 6, call 14: does the placement of single-wavefront workgroups cost the HEADLINE launch anything?  2^20 blocks, one block per lane: workgroups of one wavefront with dual
# ring stores (the product), the same with wrapped rows, workgroups of four wavefronts (wrapped rows)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call14; rm -rf $O; mkdir -p $O
cat > /tmp/wg4_headline.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from lz4net_amd import batch, _lib
for dist in (2, 3):
    n = 1 << 20
    raw = batch.synth(dist, 20260925, 0, n)
    comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
    back = torch.empty_like(raw)
    used = torch.empty(n, dtype=torch.int32, device="cuda")
    for rep in range(2):
        for name, knobs in (("one wavefront per workgroup, dual ring stores", dict(decoder_wg4=1)), ("one wavefront per workgroup, wrapped rows", dict(decoder_wg4=1, decoder_wrapped_stores=1)),
                            ("four wavefronts per workgroup, dual ring stores", dict(decoder_wg4=3)), ("four wavefronts per workgroup, wrapped rows", dict(decoder_wg4=3, decoder_wrapped_stores=1))):
            with _lib.tuning(decoder="lane", decoder_persist=2, **knobs):
                batch.decode(comp, clen, back, batch.BLOCK, result=used)
                torch.cuda.synchronize()
                ts = []
                for _ in range(4):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); batch.decode(comp, clen, back, batch.BLOCK, result=used); b.record(); b.synchronize()
                    ts.append(a.elapsed_time(b))
            ok = bool((used == clen).all()) and batch.count_mismatches(raw, back, batch.BLOCK) == 0
            print("dist %d 2^20 blocks, %-52s %8.3f ms %8.1f GB/s ok=%s" % (dist, name + ":", min(ts), n * 65536 / min(ts) / 1e6, ok), flush=True)
    del raw, comp, back
    torch.cuda.empty_cache()
PY
timeout 1200 python /tmp/wg4_headline.py 2>&1 | grep dist | tee $O/decoder_headline_workgroup_shape.txt
