"""Does the slab probe predict the lane encoder's rate?    In a fresh process shaped like bench.py (2^20-block raw / compressed / decode
buffers allocated first), the fast encode of D2 is timed with the slab built three ways in turn -- measured candidates (default), the first candidate
unmeasured, measured again -- each after lz4hip_release_workspaces(); every line carries what the probe said about the slab the encode ran on.
usage: python tools/enc_slab_probe_vs_rate.py [blocks]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
raw = batch.synth(2, 20260925, 0, n)
comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
back = torch.empty((n, batch.BLOCK), dtype=torch.uint8, device="cuda")
clen = torch.empty(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for tries in (0, 1, 0, 1):
    _lib.check(_lib.lib().lz4hip_release_workspaces())
    _lib.tuning_set("encoder_slab_tries", tries)
    t0 = time.perf_counter()
    batch.encode(raw[:16384], batch.BLOCK, comp[:16384], batch.BOUND, result=clen[:16384])
    batch.encode(raw[: 1 << 19], batch.BLOCK, comp[: 1 << 19], batch.BOUND, result=clen[: 1 << 19])      # (the full residency: the slab at its final size)
    torch.cuda.synchronize()
    setup = time.perf_counter() - t0
    ms = []
    for _ in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); batch.encode(raw, batch.BLOCK, comp, batch.BOUND, result=clen); b.record(); b.synchronize()
        ms.append(a.elapsed_time(b))
    print(f"blocks {n} slab_tries={tries or 4}: encode {n * 65536 / min(ms) / 1e6:6.2f} GB/s (passes {ms[0]:.0f} / {ms[1]:.0f} ms)  probe {_lib.tuning_get('encoder_slab_rate') / 1000:.2f} G steps/s, "
          f"{_lib.tuning_get('encoder_slab_tried')} candidate(s) built, {_lib.tuning_get('encoder_slab_chunks')} chunks, set-up + warm-up {setup:.2f} s", flush=True)
