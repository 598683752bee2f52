"""The wavefront-mapped decoder (forced) at several batch sizes, D2 and D3, with a byte check; plus the host-pointer decode of 16 384 / 4 096 blocks.
usage: python tools/wave_decode_rates.py"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lz4net_amd import batch, _lib

for dist in (2, 3):
    n = 65536
    raw = batch.synth(dist, 20260925, 0, n)
    comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
    back = torch.empty_like(raw)
    _lib.tuning_set("decoder", "wave")
    for m in (512, 4096, 16384, 65536):
        back.zero_()
        batch.decode(comp[:m], clen[:m], back[:m], batch.BLOCK)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); used = batch.decode(comp[:m], clen[:m], back[:m], batch.BLOCK); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
        ok = bool((used == clen[:m]).all()) and batch.count_mismatches(raw[:m], back[:m], batch.BLOCK) == 0
        print(f"wavefront decoder dist {dist} blocks {m:6d}: {min(ts):8.3f} ms {m * 65536 / min(ts) / 1e6:8.2f} GB/s ok={ok}", flush=True)
    _lib.tuning_set("decoder", "auto")
    if dist == 2:
        for m in (16384, 4096):
            comp_h, raw_h = comp[:m].cpu().numpy(), raw[:m].cpu().numpy()
            clen_h = clen[:m].cpu().numpy().astype(np.int32)
            back_h = np.zeros_like(raw_h); caps_h = np.full(m, batch.BLOCK, np.int32); res_h = np.zeros(m, np.int32)
            hb = _lib.Batch(src=comp_h.ctypes.data, src_off=None, src_stride=comp_h.strides[0], src_len=clen_h.ctypes.data, dst=back_h.ctypes.data, dst_off=None,
                            dst_stride=back_h.strides[0], dst_cap=caps_h.ctypes.data, dst_cap_all=0, src_len_all=0, result=res_h.ctypes.data, n_blocks=m)
            _lib.check(_lib.lib().lz4hip_decode_batch_host(C.byref(hb), 1))
            best = None
            for _ in range(4):
                t1 = time.perf_counter(); _lib.check(_lib.lib().lz4hip_decode_batch_host(C.byref(hb), 1)); dt = time.perf_counter() - t1
                best = dt if best is None else min(best, dt)
            print(f"host-pointer decode dist {dist} blocks {m}: {m * 65536 / best / 1e9:.2f} GB/s ok={bool((res_h == clen_h).all()) and bool(np.array_equal(back_h, raw_h))}", flush=True)
