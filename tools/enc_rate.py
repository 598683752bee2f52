"""Device-resident fast-encode rate per distribution (uncompressed GB/s), with a bit-exactness spot check."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from lz4net_amd import batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
for dist in (2, 3, 1, 0):
    raw = batch.synth(dist, 7, 0, n)
    comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    batch.encode(raw[:4096], batch.BLOCK, comp[:4096], batch.BOUND)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND); b.record(); b.synchronize()
        t = a.elapsed_time(b)
        best = t if best is None else min(best, t)
    back = torch.empty_like(raw)
    used = batch.decode(comp, clen, back, batch.BLOCK)
    ok = bool((used == clen).all()) and batch.count_mismatches(raw, back, batch.BLOCK) == 0
    print("dist %d blocks %d: encode %.2f GB/s ratio %.4f roundtrip %s" % (dist, n, n * 65536 / best / 1e6, float(clen.double().sum()) / (n * 65536), ok))
