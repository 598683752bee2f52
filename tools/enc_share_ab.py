"""Fast encode of a large batch: round-2 dispatch (two launches) vs the three-launch dispatch in which the blocks handed over are
shared between the lane mapping and a concurrent wavefront-mapped launch.  usage: python tools/enc_share_ab.py [blocks] [dists]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dists = [int(d) for d in (sys.argv[2] if len(sys.argv) > 2 else "2,3,1").split(",")]
for dist in dists:
    raw = batch.synth(dist, 20260925, 0, n)
    comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    clen = torch.empty(n, dtype=torch.int32, device="cuda")
    batch.encode(raw[:16384], batch.BLOCK, comp[:16384], batch.BOUND, result=clen[:16384])
    torch.cuda.synchronize()
    sums = {}
    for no_share in (1, 0, 1, 0):
        _lib.tuning_set("encoder_no_share", no_share)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); batch.encode(raw, batch.BLOCK, comp, batch.BOUND, result=clen); b.record(); b.synchronize()
        t = a.elapsed_time(b)
        sums[no_share] = (int(batch.checksum(comp, clen).sum().item()), int(clen.to(torch.int64).sum().item()))
        print(f"dist={dist} blocks={n} {'two launches (round 2)' if no_share else 'three launches (shared)'}: {n * 65536 / t / 1e6:7.2f} GB/s  {t:9.2f} ms", flush=True)
    print(f"dist={dist}: identical bytes: {sums[0] == sums[1]}", flush=True)
    del raw, comp
    torch.cuda.empty_cache()
