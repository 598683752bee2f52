"""LZ4HC lane launch at 2^18 blocks: one knob at a time around the defaults (control-flow batching, sub-chunks, residency); compressed lengths and
checksums of every setting compared with the default's.   usage: python tools/hc_knob_sweep.py [blocks] [dists]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
dists = [int(d) for d in (sys.argv[2] if len(sys.argv) > 2 else "2,3").split(",")]
SETTINGS = [{}, {"hc_ctrl_every": 4}, {"hc_ctrl_every": 16}, {"hc_ctrl_lanes": 16}, {"hc_ctrl_lanes": 48}, {"hc_ctrl_every": 16, "hc_ctrl_lanes": 48},
            {"hc_sub_chunks": 1}, {"hc_sub_chunks": 3}, {"hc_waves_per_cu": 12}, {"hc_waves_per_cu": 20}, {}]
for dist in dists:
    raw = batch.synth(dist, 7, 0, n)
    comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    batch.encode(raw, batch.BLOCK, comp, batch.BOUND, hc=True)
    torch.cuda.synchronize()
    ref = None
    for st in SETTINGS:
        with _lib.tuning(**st):
            best = None
            for _ in range(2):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND, hc=True); b.record(); b.synchronize()
                t = a.elapsed_time(b)
                best = t if best is None else min(best, t)
            sig = (clen.clone(), batch.checksum(comp, clen).clone())
        same = True if ref is None else bool((sig[0] == ref[0]).all()) and bool((sig[1] == ref[1]).all())
        ref = ref or sig
        print(f"dist {dist} blocks {n} {st or 'defaults'}: {n * 65536 / best / 1e6:.3f} GB/s ({best:.0f} ms) identical: {same}", flush=True)
