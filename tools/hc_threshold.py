"""LZ4HC: wavefront mapping vs lane mapping for small batches (where is the crossover?)."""
import os, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib
nmax = 1 << 16
dist = int(sys.argv[1]) if len(sys.argv) > 1 else 2
raw = batch.synth(dist, 3, 0, nmax)
comp = torch.empty((nmax, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
for n in (512, 1024, 2048, 4096, 8192, 16384, 32768):
    row = []
    for name in ("wave", "lane"):
        _lib.tuning_set("hc", name)
        batch.encode(raw[:256], batch.BLOCK, comp[:256], batch.BOUND, hc=True)
        torch.cuda.synchronize()
        best = None
        for _ in range(2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); batch.encode(raw[:n], batch.BLOCK, comp[:n], batch.BOUND, hc=True); b.record(); b.synchronize()
            t = a.elapsed_time(b); best = t if best is None else min(best, t)
        row.append("%s %.0f ms %.2f GB/s" % (name, best, n * 65536 / best / 1e6))
    print("n=%6d: %s" % (n, " | ".join(row)), flush=True)
