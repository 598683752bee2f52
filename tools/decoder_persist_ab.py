"""Lane decoder: one block per lane under hardware dispatch (persist=2) vs a persistent grid (persist=1) vs the automatic choice (persist=0);
the persistent grid: whose lanes pull blocks from a counter (knob
decoder_persist) -- on the homogeneous bench batches and on a heterogeneous batch (half of the blocks are zeros, which the default
dispatch routes to the wavefront mapping: the lane launch then runs wavefronts with half of their lanes idle)."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib


def rate(comp, clen, back, raw, forced):
    out = []
    for persist in (2, 1, 0):
        _lib.tuning_set("decoder_persist", persist)
        _lib.tuning_set("decoder", "lane" if forced else "auto")
        back.zero_()
        used = batch.decode(comp, clen, back, batch.BLOCK)
        torch.cuda.synchronize()
        ok = bool((used == clen).all()) and batch.count_mismatches(raw, back, batch.BLOCK) == 0
        best = None
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); batch.decode(comp, clen, back, batch.BLOCK); b.record(); b.synchronize()
            t = a.elapsed_time(b); best = t if best is None else min(best, t)
        out.append("persist=%d %.2f ms %.1f GB/s ok=%s" % (persist, best, comp.shape[0] * 65536 / best / 1e6, ok))
    _lib.tuning_set("decoder_persist", 0); _lib.tuning_set("decoder", "auto")
    return " | ".join(out)


for n in (1 << 20, 1 << 18, 1 << 16):
    for dist in (2, 3):
        raw = batch.synth(dist, 20260925, 0, n)
        comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
        clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
        back = torch.empty_like(raw)
        print("homogeneous dist %d n=%7d (lane mapping forced): %s" % (dist, n, rate(comp, clen, back, raw, True)), flush=True)
        if dist == 2:
            # heterogeneous: every second block is zeros (default dispatch: those go to the wavefront launch)
            z = batch.synth(0, 1, 0, n // 2)
            raw[1::2] = z
            clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
            print("mixed D2 / zeros    n=%7d (default dispatch)    : %s" % (n, rate(comp, clen, back, raw, False)), flush=True)
            del z
        del raw, comp, back
        torch.cuda.empty_cache()
