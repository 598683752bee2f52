"""Decode rate of the wavefront mapping vs the lane mapping (default generation) for small and mid-size batches (where is the crossover?).
usage: python tools/decode_threshold.py [dists] [sizes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib
dists = [int(d) for d in (sys.argv[1] if len(sys.argv) > 1 else "2,3").split(",")]
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1024,2048,4096,8192,16384,32768,65536,131072,262144").split(",")]
for dist in dists:
    nmax = max(sizes)
    raw = batch.synth(dist, 3, 0, nmax)
    comp = torch.empty((nmax, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
    back = torch.empty_like(raw)
    for n in sizes:
        row = []
        for name, gen in (("wave", 0), ("lane", 0)):
            _lib.tuning_set("decoder", name)
            _lib.tuning_set("decoder_gen", gen)
            batch.decode(comp[:n], clen[:n], back[:n], batch.BLOCK)
            torch.cuda.synchronize()
            best = None
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); batch.decode(comp[:n], clen[:n], back[:n], batch.BLOCK); b.record(); b.synchronize()
                t = a.elapsed_time(b); best = t if best is None else min(best, t)
            row.append("%s%s %.2f ms %.1f GB/s" % (name, gen or "", best, n * 65536 / best / 1e6))
        print("dist %d n=%6d: %s" % (dist, n, " | ".join(row)), flush=True)
