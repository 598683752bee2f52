"""Timeline of ONE host-pointer decode (lz4hip_decode_batch_host, D2, pageable caller arrays) for rocprofv3 --kernel-trace --memory-copy-trace:
   python tools/host_decode_timeline.py [blocks] ; the third decode call is the one to look at (tools/host_timeline_report.py)."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: F401  (HIP runtime first)
from lz4net_amd import _lib, batch

m = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dist = int(sys.argv[2]) if len(sys.argv) > 2 else 2
raw_d = batch.synth(dist, 1, 0, m)
raw_h = raw_d.cpu().numpy()
comp_h = np.zeros((m, batch.BOUND_STRIDE), np.uint8)
lens = np.full(m, batch.BLOCK, np.int32)
caps = np.full(m, batch.BOUND, np.int32)
clen = np.zeros(m, np.int32)
eb = _lib.Batch(src=raw_h.ctypes.data, src_off=None, src_stride=raw_h.strides[0], src_len=lens.ctypes.data,
                dst=comp_h.ctypes.data, dst_off=None, dst_stride=comp_h.strides[0], dst_cap=caps.ctypes.data,
                dst_cap_all=0, src_len_all=0, result=clen.ctypes.data, n_blocks=m)
_lib.check(_lib.lib().lz4hip_encode_batch_host(C.byref(eb), 0))
back_h = np.zeros_like(raw_h)
res = np.zeros(m, np.int32)
db = _lib.Batch(src=comp_h.ctypes.data, src_off=None, src_stride=comp_h.strides[0], src_len=clen.ctypes.data,
                dst=back_h.ctypes.data, dst_off=None, dst_stride=back_h.strides[0], dst_cap=lens.ctypes.data,
                dst_cap_all=0, src_len_all=0, result=res.ctypes.data, n_blocks=m)
for k in range(3):
    t = time.perf_counter(); _lib.check(_lib.lib().lz4hip_decode_batch_host(C.byref(db), 1)); dt = time.perf_counter() - t
    print("decode call %d: %.2f ms = %.2f GB/s" % (k, dt * 1e3, m * 65536 / dt / 1e9), flush=True)
    time.sleep(0.05)
print("ok=%s" % (bool((res == clen).all()) and bool(np.array_equal(back_h, raw_h))))
