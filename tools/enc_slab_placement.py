"""Is the fast encoder's bimodal rate (44-46 vs 53-54 GB/s on D2) a property of WHERE its table slab was allocated?  One process, one resident
batch: encode, give the slab back (lz4hip_release_workspaces), encode again (a new hipMalloc), ... and between some of the rounds allocate /
free torch tensors of odd sizes so that the next slab lands elsewhere."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
raw = batch.synth(2, 20260925, 0, n)
comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
back = torch.empty_like(raw)          # (the bench holds three buffers of this size)
junk = []
for r in range(rounds):
    ts = []
    for _ in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    free = torch.cuda.mem_get_info()[0]
    print("round %d: encode %.2f / %.2f GB/s (first pass allocates the slab), free %.1f GiB, junk tensors %d" % (r, n * 65536 / ts[0] / 1e6, n * 65536 / ts[1] / 1e6, free / 2**30, len(junk)), flush=True)
    _lib.check(_lib.lib().lz4hip_release_workspaces())
    if r % 2 == 1:
        junk.append(torch.empty((3 << 30) + (r << 21) + 4096 * r, dtype=torch.uint8, device="cuda"))
