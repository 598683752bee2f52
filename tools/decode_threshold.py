"""Decode rate of the wavefront mapping vs the lane mapping for small batches (where is the crossover?)."""
import os, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib
for dist in (2, 3):
    nmax = 1 << 16
    raw = batch.synth(dist, 3, 0, nmax)
    comp = torch.empty((nmax, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
    back = torch.empty_like(raw)
    for n in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
        row = []
        for name in ("wave", "lane"):
            _lib.tuning_set("decoder", name)
            batch.decode(comp[:n], clen[:n], back[:n], batch.BLOCK)
            torch.cuda.synchronize()
            best = None
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); batch.decode(comp[:n], clen[:n], back[:n], batch.BLOCK); b.record(); b.synchronize()
                t = a.elapsed_time(b); best = t if best is None else min(best, t)
            row.append("%s %.2f ms %.1f GB/s" % (name, best, n * 65536 / best / 1e6))
        print("dist %d n=%6d: %s" % (dist, n, " | ".join(row)), flush=True)
