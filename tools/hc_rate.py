"""Device-resident LZ4HC encode rate for a few residency settings (uncompressed GB/s), with a round-trip check."""
import os
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
settings = sys.argv[2].split(",") if len(sys.argv) > 2 else ["4", "8", "16", "20"]
for dist in (2, 3):
    raw = batch.synth(dist, 7, 0, n)
    comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    for wpc in settings:
        _lib.tuning_set("hc_waves_per_cu", int(wpc))
        batch.encode(raw[:4096], batch.BLOCK, comp[:4096], batch.BOUND, hc=True)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND, hc=True); b.record(); b.synchronize()
        t = a.elapsed_time(b)
        back = torch.empty_like(raw)
        used = batch.decode(comp, clen, back, batch.BLOCK)
        ok = bool((used == clen).all()) and batch.count_mismatches(raw, back, batch.BLOCK) == 0
        print("dist %d blocks %d waves/CU %s: HC encode %.3f GB/s ratio %.4f roundtrip %s" % (dist, n, wpc, n * 65536 / t / 1e6, float(clen.double().sum()) / (n * 65536), ok), flush=True)
        del back
