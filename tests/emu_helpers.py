"""ctypes front-end for tests/simt/libsimt_kernels.so (TEST INFRASTRUCTURE): runs the real kernel
sources under the CPU SIMT emulator.  Same call shapes as the GPU batch API so tests can share code."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "simt"))
from build_emu import build  # noqa: E402

_lib = None
_starved = None
_use_starved = False


def lib():
    global _lib, _starved
    if _use_starved:
        if _starved is None:
            _starved = C.CDLL(build(starved=True))
        return _starved
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.emu_compare.restype = C.c_ulonglong
        _lib.emu_steps.restype = C.c_ulonglong
    return _lib


class starved_flush:
    """with emu.starved_flush(): ...  -- run the kernels of the 'starved' build (4 flush records per round)."""

    def __enter__(self):
        global _use_starved
        _use_starved = True

    def __exit__(self, *a):
        global _use_starved
        _use_starved = False
        return False


def _p(a):
    return C.c_void_p(a.ctypes.data)


def pack(rows, pad=16):
    n = len(rows)
    stride = max([len(r) for r in rows] + [1]) + pad
    buf = np.zeros((n, stride), np.uint8)
    for i, r in enumerate(rows):
        buf[i, :len(r)] = r
    return buf, np.array([len(r) for r in rows], np.int32)


def decode(comps, out_sizes, known=True, src_lens=None, waves_per_group=1, auto=False, lane=0, stage=64, gen=2):
    src, sl = pack(comps)
    if src_lens is not None:
        sl = np.array(src_lens, np.int32)
    caps = np.array(out_sizes, np.int32)
    ds = max(int(caps.max()), 1) + 64
    dst = np.full((len(comps), ds), 0xA5, np.uint8)
    res = np.full(len(comps), -12345678, np.int32)
    args = (int(known), _p(src), C.c_int64(src.shape[1]), _p(sl), _p(dst), C.c_int64(ds), _p(caps), _p(res), C.c_int64(len(comps)))
    if lane and gen == 6:                       # generation 4 in workgroups of four wavefronts (lane = 1: dual ring stores, 2: wrapped rows)
        lib().emu_decode_lane4_wg4(*args, 0, int(lane == 2))
    elif lane and gen == 5:                     # the persistent variant of generation 4: `lane` wavefronts in the grid
        lib().emu_decode_lane4_persistent(*args, 0, lane)
    elif lane and gen == 4:
        lib().emu_decode_lane4(*args, 0, lane)
    elif lane and gen == 3:
        lib().emu_decode_lane3(*args, 0, lane, stage)
    elif lane:
        lib().emu_decode_lane(*args, 0, lane, stage)
    elif auto:        # the library's default: the batch is partitioned between the two mappings
        lib().emu_decode_lane4(*args, 2, 59192)
        lib().emu_decode(*args, waves_per_group, 1)
    else:
        lib().emu_decode(*args, waves_per_group, 0)
    return res, dst


def encode(blocks, caps=None, hc=False, groups=2, lane=False, conv=False, nat=False, lcp=False, wg5=False):
    src, sl = pack(blocks)
    if caps is None:
        caps = [len(b) + len(b) // 255 + 16 for b in blocks]
    caps = np.array(caps, np.int32)
    ds = max(int(caps.max()), 1) + 64
    dst = np.full((len(blocks), ds), 0xA5, np.uint8)
    res = np.zeros(len(blocks), np.int32)
    args = (_p(src), C.c_int64(src.shape[1]), _p(sl), _p(dst), C.c_int64(ds), _p(caps), _p(res), C.c_int64(len(blocks)))
    if hc and lcp:
        lib().emu_encode_hc_lcp(*args, groups)
    elif hc and nat:
        lib().emu_encode_hc_nat(*args, groups)
    elif hc and conv:
        lib().emu_encode_hc_conv(*args, 1, int(max(len(b) for b in blocks) > 65536))
    elif hc and lane:
        lib().emu_encode_hc_lane(*args, 1, int(max(len(b) for b in blocks) > 65536))
    elif hc:
        lib().emu_encode_hc(*args, groups, int(max(len(b) for b in blocks) > 65536))
    elif lane:
        lib().emu_encode_fast_lane(*args, 1)
    elif wg5:
        lib().emu_encode_fast_wg5(*args)
    else:
        lib().emu_encode_fast(*args)
    return res, dst


def encode_hc_lane_static(blocks):
    """LZ4HC lane kernel's per-block function with STATIC assignment: lane L encodes blocks L, L+64, L+128, ... in order on
    one slab (slab-reuse test)."""
    src, sl = pack(blocks)
    caps = np.array([len(b) + len(b) // 255 + 16 for b in blocks], np.int32)
    ds = max(int(caps.max()), 1) + 64
    dst = np.full((len(blocks), ds), 0xA5, np.uint8)
    res = np.zeros(len(blocks), np.int32)
    lib().emu_encode_hc_lane_static(_p(src), C.c_int64(src.shape[1]), _p(sl), _p(dst), C.c_int64(ds), _p(caps), _p(res), C.c_int64(len(blocks)))
    return res, dst


def encode_two_launches(blocks, caps=None):
    """launch_encode's default for large batches: wavefront-per-block launch with the hand-over rule, then the lane-per-block
    launch over the blocks handed over.  Returns (result, dst, deferred flags)."""
    src, sl = pack(blocks)
    if caps is None:
        caps = [len(b) + len(b) // 255 + 16 for b in blocks]
    caps = np.array(caps, np.int32)
    ds = max(int(caps.max()), 1) + 64
    dst = np.full((len(blocks), ds), 0xA5, np.uint8)
    res = np.zeros(len(blocks), np.int32)
    deferred = np.zeros(len(blocks), np.int32)
    lib().emu_encode_fast_two_launches(_p(src), C.c_int64(src.shape[1]), _p(sl), _p(dst), C.c_int64(ds), _p(caps), _p(res),
                                       C.c_int64(len(blocks)), _p(deferred))
    return res, dst, deferred


def synth(dist, seed, first_block, n, length, stride=None, block_step=1):
    stride = length if stride is None else stride
    out = np.zeros((n, max(stride, 1)), np.uint8)
    lib().emu_synth(dist, C.c_uint64(seed), C.c_uint64(first_block), C.c_uint64(block_step), C.c_int64(n), _p(out), C.c_int64(out.shape[1]), length)
    return out


def checksum(rows):
    buf, ln = pack(rows, pad=0)
    sums = np.zeros(len(rows), np.uint64)
    lib().emu_checksum(_p(buf), C.c_int64(buf.shape[1]), _p(ln), _p(sums), C.c_int64(len(rows)))
    return sums


def compare(a, b, lens):
    lens = np.array(lens, np.int32)
    return lib().emu_compare(_p(a), C.c_int64(a.shape[1]), _p(b), C.c_int64(b.shape[1]), _p(lens), C.c_int64(a.shape[0]))
