"""The wavefront-mapped fast encoder (forced) at several batch sizes, D2 and D3: kernel time by HIP events (best of 3), compressed lengths and checksums
printed as one number so that two library variants can be compared; plus the host-pointer fast encode of 16 384 / 4 096 blocks.
usage: python tools/enc_wave_rates.py"""
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
    _lib.tuning_set("encoder", "wave")
    for m in (512, 2560, 4096, 16384, 65536):
        clen = batch.encode(raw[:m], batch.BLOCK, comp[:m], batch.BOUND)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); clen = batch.encode(raw[:m], batch.BLOCK, comp[:m], batch.BOUND); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
        digest = int(batch.checksum(comp[:m], clen).to(torch.int64).sum().item()) & 0xFFFFFFFFFFFF
        print(f"wavefront encoder dist {dist} blocks {m:6d}: {min(ts):8.3f} ms {m * 65536 / min(ts) / 1e6:8.2f} GB/s  bytes {int(clen.sum())} digest {digest:012x}", flush=True)
    _lib.tuning_set("encoder", "auto")
    if dist == 2:
        for m in (16384, 4096):
            raw_h = raw[:m].cpu().numpy()
            enc_h = np.zeros((m, batch.BOUND_STRIDE), np.uint8)
            ecap_h = np.full(m, batch.BOUND, np.int32); elen_h = np.full(m, batch.BLOCK, np.int32); eres_h = np.zeros(m, np.int32)
            eb = _lib.Batch(src=raw_h.ctypes.data, src_off=None, src_stride=raw_h.strides[0], src_len=elen_h.ctypes.data, dst=enc_h.ctypes.data, dst_off=None,
                            dst_stride=enc_h.strides[0], dst_cap=ecap_h.ctypes.data, dst_cap_all=0, src_len_all=batch.BLOCK, result=eres_h.ctypes.data, n_blocks=m)
            _lib.check(_lib.lib().lz4hip_encode_batch_host(C.byref(eb), 0))
            best = None
            for _ in range(4):
                t1 = time.perf_counter(); _lib.check(_lib.lib().lz4hip_encode_batch_host(C.byref(eb), 0)); dt = time.perf_counter() - t1
                best = dt if best is None else min(best, dt)
            print(f"host-pointer fast encode dist {dist} blocks {m}: {m * 65536 / best / 1e9:.2f} GB/s  bytes {int(eres_h.sum())}", flush=True)
