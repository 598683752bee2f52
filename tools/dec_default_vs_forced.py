"""Default decoder dispatch against the forced mappings at mid batch sizes, same buffers (what does the default path add to the lane kernel's own time?).
usage: python tools/dec_default_vs_forced.py [dist]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib

dist = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nmax = 262144
raw = batch.synth(dist, 20260925, 0, nmax)
comp = torch.empty((nmax, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
back = torch.empty_like(raw)


def timed(fn, reps=5):
    best = None
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        t = a.elapsed_time(b)
        best = t if best is None else min(best, t)
    return best


for m in (8192, 16384, 32768, 65536, 131072, 262144):
    row = []
    for name, knobs in (("default", {}), ("lane, one block per lane", dict(decoder="lane", decoder_persist=2)), ("lane, persistent", dict(decoder="lane", decoder_persist=1)),
                        ("lane, automatic form", dict(decoder="lane")), ("wave", dict(decoder="wave"))):
        with _lib.tuning(**knobs):
            batch.decode(comp[:m], clen[:m], back[:m], batch.BLOCK)
            torch.cuda.synchronize()
            t = timed(lambda: batch.decode(comp[:m], clen[:m], back[:m], batch.BLOCK))
        row.append("%s %7.3f ms %6.1f GB/s" % (name, t, m * 65536 / t / 1e6))
    print("dist %d blocks %7d: %s" % (dist, m, " | ".join(row)), flush=True)
