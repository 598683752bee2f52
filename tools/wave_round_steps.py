"""Where the rounds of the two wavefront-mapped kernels begin: kernel time (HIP events, best of 3) of the forced wavefront encoder and decoder on D2 at
batch sizes of 1 ... 12 blocks per CU (256 CUs), one line per size.  A step in the time between k and k+1 blocks per CU = k workgroups resident per CU.
usage: python tools/wave_round_steps.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib

n = 4096
raw = batch.synth(2, 20260925, 0, n)
comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
out = torch.empty_like(raw)
_lib.tuning_set("encoder", "wave")
clen_all = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
torch.cuda.synchronize()


def best(fn):
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts)


for per_cu in (1, 2, 4, 6, 8, 9, 10, 11, 12, 14, 16):
    m = 256 * per_cu
    te = best(lambda: batch.encode(raw[:m], batch.BLOCK, comp[:m], batch.BOUND))
    _lib.tuning_set("decoder", "wave")
    td = best(lambda: batch.decode(comp[:m], clen_all[:m], out[:m], batch.BLOCK))
    _lib.tuning_set("decoder", "auto")
    print(f"{per_cu:3d} blocks per CU ({m:5d} blocks): wavefront encoder {te:7.3f} ms, wavefront decoder {td:7.3f} ms", flush=True)
_lib.tuning_set("encoder", "auto")
