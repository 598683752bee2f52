"""The two wavefront-mapped kernels on a batch that takes them alone: one timed fast encode and one timed decode of N fuzzer-style blocks
(what tools/pmc_wave_small_batch.sh profiles).  usage: python tools/wave_small_batch.py [blocks] [dist]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dist = int(sys.argv[2]) if len(sys.argv) > 2 else 2
raw = batch.synth(dist, 20260925, 0, n)
comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
back = torch.empty_like(raw)
_lib.tuning_set("encoder", "wave"); _lib.tuning_set("decoder", "wave")
clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
batch.decode(comp, clen, back, batch.BLOCK)
torch.cuda.synchronize()
for name, fn in (("encode", lambda: batch.encode(raw, batch.BLOCK, comp, batch.BOUND)), ("decode", lambda: batch.decode(comp, clen, back, batch.BLOCK))):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); b.synchronize()
    print(f"{name}: dist {dist} blocks {n}: {a.elapsed_time(b):.3f} ms", flush=True)
assert batch.count_mismatches(raw, back, batch.BLOCK) == 0
