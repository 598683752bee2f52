"""Wavefront-mapped fast encoder, first version (encode_fast_block<false>) vs second (encode_fast_block64k), same process, same batches
(a library built with LZ4HIP_BUILD_FLAGS=-DLZ4HIP_TUNING_BUILD holds both; knob encoder_wave_version): kernel time by HIP events, best
of 3, and the compressed bytes compared.   usage: python tools/enc_wave_versions_ab.py [dists] [batch sizes]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib

dists = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2,3,1,0").split(",")]
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "512,2560,4096,16384,65536").split(",")]
assert "+tuning" in _lib.lib().lz4hip_build_id().decode(), "needs a tuning build (both versions)"
_lib.tuning_set("encoder", "wave")


def timed(fn):
    best = None
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        t = a.elapsed_time(b)
        best = t if best is None else min(best, t)
    return best


for dist in dists:
    nmax = max(sizes)
    raw = batch.synth(dist, 20260925, 0, nmax)
    comp = {v: torch.empty((nmax, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda") for v in (1, 2)}
    for n in sizes:
        out = {}
        for v in (1, 2):
            _lib.tuning_set("encoder_wave_version", v)
            clen = batch.encode(raw[:n], batch.BLOCK, comp[v][:n], batch.BOUND)
            torch.cuda.synchronize()
            ms = timed(lambda: batch.encode(raw[:n], batch.BLOCK, comp[v][:n], batch.BOUND))
            out[v] = (ms, clen, batch.checksum(comp[v][:n], clen))
        same = bool((out[1][1] == out[2][1]).all()) and bool((out[1][2] == out[2][2]).all())
        print("dist %d blocks %6d: first version %8.2f ms %7.2f GB/s | second %8.2f ms %7.2f GB/s | x%.2f | bytes equal %s" % (
            dist, n, out[1][0], n * 65536 / out[1][0] / 1e6, out[2][0], n * 65536 / out[2][0] / 1e6, out[1][0] / out[2][0], same), flush=True)
_lib.tuning_set("encoder_wave_version", 0)
