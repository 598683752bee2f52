"""(Experiment NOT kept: needs tools/ab/encoder_shared_handover.patch applied -- profiles/r06/encoder_shared_handover_ab.txt.)  Fast encode of a large batch: the blocks the first launch hands over go to the lane-per-block grid alone (knob encoder_share = 1, rounds 2-5)
or are shared between it and a persistent wavefront grid working from the back of the batch (0, default).  Same process, same batches:
kernel time by HIP events (best of 2), compressed lengths and checksums compared.
usage: python tools/enc_shared_handover_ab.py [dists] [batch sizes]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib

dists = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2,3,1,0").split(",")]
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "65536,262144,1048576").split(",")]


def timed(fn, reps=2):
    best = None
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        t = a.elapsed_time(b)
        best = t if best is None else min(best, t)
    return best


for dist in dists:
    nmax = max(sizes)
    raw = batch.synth(dist, 20260925, 0, nmax)
    comp = torch.empty((nmax, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    for n in sizes:
        out = {}
        for share in (1, 0):
            _lib.tuning_set("encoder_share", share)
            clen = batch.encode(raw[:n], batch.BLOCK, comp[:n], batch.BOUND)
            torch.cuda.synchronize()
            ms = timed(lambda: batch.encode(raw[:n], batch.BLOCK, comp[:n], batch.BOUND))
            out[share] = (ms, clen.clone(), batch.checksum(comp[:n], clen).clone())
        same = bool((out[0][1] == out[1][1]).all()) and bool((out[0][2] == out[1][2]).all()) and bool((out[0][1] > 0).all())
        print("dist %d blocks %8d: lane grid alone %9.2f ms %7.2f GB/s | shared with the wavefront grid %9.2f ms %7.2f GB/s | x%.2f | bytes equal %s" % (
            dist, n, out[1][0], n * 65536 / out[1][0] / 1e6, out[0][0], n * 65536 / out[0][0] / 1e6, out[1][0] / out[0][0], same), flush=True)
    del raw, comp
    torch.cuda.empty_cache()
_lib.tuning_set("encoder_share", 0)
