"""LZ4HC lane launch: sub-chunks whose table builders and lane kernels overlap (knob hc_sub_chunks) vs one after the other.
Same box, same batch, every setting's compressed lengths + checksums compared with the first one's."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
settings = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
for dist in (2, 3):
    raw = batch.synth(dist, 7, 0, n)
    comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    batch.encode(raw[:4096], batch.BLOCK, comp[:4096], batch.BOUND, hc=True)
    ref = None
    for subs in settings:
        _lib.tuning_set("hc_sub_chunks", subs)
        best = None
        for _ in range(2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND, hc=True); b.record(); b.synchronize()
            t = a.elapsed_time(b)
            best = t if best is None else min(best, t)
        sums = batch.checksum(comp, clen)
        sig = (clen.clone(), sums.clone())
        same = True if ref is None else bool((sig[0] == ref[0]).all()) and bool((sig[1] == ref[1]).all())
        if ref is None:
            ref = sig
        print("dist %d blocks %d sub-chunks %d: HC encode %.3f GB/s (%.1f ms) identical to the first setting: %s" % (dist, n, subs, n * 65536 / best / 1e6, best, same), flush=True)
    _lib.tuning_set("hc_sub_chunks", 0)
    back = torch.empty_like(raw)
    used = batch.decode(comp, clen, back, batch.BLOCK)
    print("  round trip:", bool((used == clen).all()) and batch.count_mismatches(raw, back, batch.BLOCK) == 0, flush=True)
    del raw, comp, back
    torch.cuda.empty_cache()
