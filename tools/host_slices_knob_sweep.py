"""PCIe-inclusive decode/encode rate of the host-pointer batch entry points for several batch sizes and `host_slices` settings."""
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from lz4net_amd import _lib, batch
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1024,4096,16384").split(",")]
slices = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3,4,6,8").split(",")]
dist = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mmax = max(sizes)
raw_d = batch.synth(dist, 1, 0, mmax)
raw_all = raw_d.cpu().numpy()
for m in sizes:
    raw_h = raw_all[:m]
    comp_h = np.zeros((m, batch.BOUND_STRIDE), np.uint8)
    lens = np.full(m, batch.BLOCK, np.int32); caps = np.full(m, batch.BOUND, np.int32); clen = np.zeros(m, np.int32)
    eb = _lib.Batch(src=raw_h.ctypes.data, src_off=None, src_stride=raw_h.strides[0], src_len=lens.ctypes.data,
                    dst=comp_h.ctypes.data, dst_off=None, dst_stride=comp_h.strides[0], dst_cap=caps.ctypes.data,
                    dst_cap_all=0, src_len_all=0, result=clen.ctypes.data, n_blocks=m)
    back_h = np.zeros_like(raw_h); res = np.zeros(m, np.int32)
    db = _lib.Batch(src=comp_h.ctypes.data, src_off=None, src_stride=comp_h.strides[0], src_len=clen.ctypes.data,
                    dst=back_h.ctypes.data, dst_off=None, dst_stride=back_h.strides[0], dst_cap=lens.ctypes.data,
                    dst_cap_all=0, src_len_all=0, result=res.ctypes.data, n_blocks=m)
    for sl in slices:
        _lib.tuning_set("host_slices", sl)
        te = td = None
        for _ in range(3):
            t = time.perf_counter(); _lib.check(_lib.lib().lz4hip_encode_batch_host(C.byref(eb), 0)); dt = time.perf_counter() - t
            te = dt if te is None else min(te, dt)
            t = time.perf_counter(); _lib.check(_lib.lib().lz4hip_decode_batch_host(C.byref(db), 1)); dt = time.perf_counter() - t
            td = dt if td is None else min(td, dt)
        ok = bool((res == clen).all()) and bool(np.array_equal(back_h, raw_h))
        print("dist %d blocks %6d slices %d: host-pointer encode %6.2f GB/s (%6.1f ms), decode %6.2f GB/s (%6.1f ms) ok=%s"
              % (dist, m, sl, m * 65536 / te / 1e9, te * 1e3, m * 65536 / td / 1e9, td * 1e3, ok), flush=True)
