"""Fast encoder (two launches) at 2^20 blocks vs the lane kernel's residency knob; LZ4HC (generation 4) vs residency and control batching.
usage: python tools/enc_residency_ab.py [fast_blocks] [hc_blocks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib

def rate(fn, n, reps=2):
    best = None
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        t = a.elapsed_time(b); best = t if best is None else min(best, t)
    return n * 65536 / best / 1e6, best

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
nh = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
for dist in (2, 3):
    raw = batch.synth(dist, 7, 0, nf)
    comp = torch.empty((nf, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    ref = None
    for wpc in (16, 24, 32):
        _lib.tuning_set("encoder_waves_per_cu", wpc)
        batch.encode(raw[:32768], batch.BLOCK, comp[:32768], batch.BOUND); torch.cuda.synchronize()
        holder = {}
        r, ms = rate(lambda: holder.__setitem__("c", batch.encode(raw, batch.BLOCK, comp, batch.BOUND)), nf)
        sig = int(batch.checksum(comp, holder["c"]).sum().item())
        ref = sig if ref is None else ref
        print(f"fast dist={dist} blocks={nf} encoder_waves_per_cu={wpc}: {r:7.2f} GB/s {ms:8.1f} ms same bytes: {sig == ref}", flush=True)
    _lib.tuning_set("encoder_waves_per_cu", 0)
    del raw, comp; torch.cuda.empty_cache()
raw = batch.synth(2, 7, 0, nh)
comp = torch.empty((nh, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
ref = None
for wpc, every, lanes in ((16, 0, 0), (12, 0, 0), (16, 16, 32), (16, 8, 48), (16, 16, 48)):
    _lib.tuning_set("hc_waves_per_cu", wpc); _lib.tuning_set("hc_ctrl_every", every); _lib.tuning_set("hc_ctrl_lanes", lanes)
    batch.encode(raw[:16384], batch.BLOCK, comp[:16384], batch.BOUND, hc=True); torch.cuda.synchronize()
    holder = {}
    r, ms = rate(lambda: holder.__setitem__("c", batch.encode(raw, batch.BLOCK, comp, batch.BOUND, hc=True)), nh)
    sig = int(batch.checksum(comp, holder["c"]).sum().item())
    ref = sig if ref is None else ref
    print(f"LZ4HC dist=2 blocks={nh} hc_waves_per_cu={wpc} ctrl={every}/{lanes}: {r:7.2f} GB/s {ms:8.1f} ms same bytes: {sig == ref}", flush=True)
