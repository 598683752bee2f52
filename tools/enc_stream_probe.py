"""Is the fast encoder's bimodal rate (44-46 vs 53-54 GB/s on D2) a property of the QUEUE the launches go to?  One process, one resident
batch, one slab: the same encode on torch's default stream and on several new streams (each maps to a hardware queue of its own), a few
timings each; the decode of the same batch beside it for contrast."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from lz4net_amd import batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
n_streams = int(sys.argv[2]) if len(sys.argv) > 2 else 6
raw = batch.synth(2, 20260925, 0, n)
comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
back = torch.empty_like(raw)
clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
torch.cuda.synchronize()
streams = [None] + [torch.cuda.Stream(priority=p) for p in ([0] * (n_streams - 2) + [-1])]
for rnd in range(2):
    for i, s in enumerate(streams):
        ctx = torch.cuda.stream(s) if s is not None else torch.cuda.stream(torch.cuda.default_stream())
        with ctx:
            te, td = [], []
            for _ in range(2):
                a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                a.record(); batch.encode(raw, batch.BLOCK, comp, batch.BOUND); b.record()
                batch.decode(comp, clen, back, batch.BLOCK); c.record(); c.synchronize()
                te.append(a.elapsed_time(b)); td.append(b.elapsed_time(c))
        print("round %d stream %d (%s): encode %.2f / %.2f GB/s   decode %.1f / %.1f GB/s" % (
            rnd, i, "default" if s is None else "new, priority %d" % s.priority, n * 65536 / te[0] / 1e6, n * 65536 / te[1] / 1e6,
            n * 65536 / td[0] / 1e6, n * 65536 / td[1] / 1e6), flush=True)
