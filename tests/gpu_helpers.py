"""numpy front-end over the HOST-pointer batch entry points of liblz4hip.so, used by the -m gpu tests.
Everything goes through the C ABI (include/lz4hip.h)."""
import ctypes as C

import numpy as np

from lz4net_amd import _lib


def pack(rows, pad=16):
    n = len(rows)
    stride = max([len(r) for r in rows] + [1]) + pad
    buf = np.zeros((n, stride), np.uint8)
    for i, r in enumerate(rows):
        buf[i, :len(r)] = r
    return buf, np.array([len(r) for r in rows], np.int32)


def _batch(src, sl, dst, caps, res):
    return _lib.Batch(src=src.ctypes.data, src_off=None, src_stride=src.strides[0], src_len=sl.ctypes.data,
                      dst=dst.ctypes.data, dst_off=None, dst_stride=dst.strides[0], dst_cap=caps.ctypes.data,
                      dst_cap_all=0, src_len_all=0, result=res.ctypes.data, n_blocks=src.shape[0])


def encode(blocks, caps=None, hc=False, canary=64, device_mask=None):
    src, sl = pack(blocks)
    if caps is None:
        caps = [len(b) + len(b) // 255 + 16 for b in blocks]
    caps = np.array(caps, np.int32)
    dst = np.full((len(blocks), max(int(caps.max()), 1) + canary), 0xA5, np.uint8)
    res = np.zeros(len(blocks), np.int32)
    b = _batch(src, sl, dst, caps, res)
    if device_mask is None:
        _lib.check(_lib.lib().lz4hip_encode_batch_host(C.byref(b), 1 if hc else 0))
    else:
        _lib.check(_lib.lib().lz4hip_encode_batch_host_multi(C.byref(b), 1 if hc else 0, device_mask))
    return res, dst


def decode(comps, out_sizes, known=True, src_lens=None, canary=64, device_mask=None):
    src, sl = pack(comps)
    if src_lens is not None:
        sl = np.array(src_lens, np.int32)
    caps = np.array(out_sizes, np.int32)
    dst = np.full((len(comps), max(int(caps.max()), 1) + canary), 0xA5, np.uint8)
    res = np.zeros(len(comps), np.int32)
    b = _batch(src, sl, dst, caps, res)
    if device_mask is None:
        _lib.check(_lib.lib().lz4hip_decode_batch_host(C.byref(b), 1 if known else 0))
    else:
        _lib.check(_lib.lib().lz4hip_decode_batch_host_multi(C.byref(b), 1 if known else 0, device_mask))
    return res, dst
