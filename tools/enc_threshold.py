"""Fast encode of mid-size batches: the wavefront mapping alone against the two-launch dispatch (wavefront launch with hand-over + lane launch): where is the
crossover?  (kLaneEncodeMinBlocks in lz4hip_api.hip.)   usage: python tools/enc_threshold.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib

for dist in (2, 3):
    raw = batch.synth(dist, 20260925, 0, 65536)
    comp = torch.empty((65536, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    with _lib.tuning(encoder="lane"):
        batch.encode(raw, batch.BLOCK, comp, batch.BOUND)          # (the slab at its final size)
    torch.cuda.synchronize()
    for n in (16384, 24576, 32768, 40960, 49152, 57344, 65536):
        out = {}
        for name in ("wave", "lane"):
            _lib.tuning_set("encoder", name)
            batch.encode(raw[:n], batch.BLOCK, comp[:n], batch.BOUND)
            torch.cuda.synchronize()
            ts = []
            for _ in range(2):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); batch.encode(raw[:n], batch.BLOCK, comp[:n], batch.BOUND); b.record(); b.synchronize()
                ts.append(a.elapsed_time(b))
            out[name] = min(ts)
        print(f"dist {dist} blocks {n:6d}: wavefront mapping alone {out['wave']:7.1f} ms, lane mapping {out['lane']:7.1f} ms", flush=True)
    _lib.tuning_set("encoder", "auto")
