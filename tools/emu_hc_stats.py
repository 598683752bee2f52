"""Lane-step accounting of the LZ4HC lane kernel (lz4hip_hc_lcp.hpp) under the SIMT emulator (TEST INFRASTRUCTURE): which state the lanes' memory steps
are spent in, and how the walk's equal case F(c) == lcp[c] is decided.    usage: python tools/emu_hc_stats.py [dist] [blocks]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu_helpers as emu
from oracle.oracle import Oracle

NAMES = ["lane-steps (lane not finished)", "  head: entry of the search position", "  hop: candidate evaluated, its entry read", "  compare: 16 bytes of both sides", "  backward extension",
         "  repeat fill", "  waiting for the batched control flow", "walk advances to the next candidate", "  equal case (or both >= 255)", "    below the cap", "    search position's byte in the register window",
         "    decided without a compare",
         "searches (walks that ended)", "  candidates evaluated by them", "  16-byte reads a bucket-contiguous table would need (own entry + candidates, 4 per read)", "  32-byte reads (8 per read)",
         "  searches with <= 1 candidate", "  2-3", "  4-7", "  8-31", "  >= 32"]
dist = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 64
o = Oracle()
blocks = [o.gen(dist, 99, i, 1)[0] for i in range(nblk)]
lib = emu.lib()
st = (C.c_ulonglong * 32)()
lib.emu_stats(st, 1)
res, dst = emu.encode(blocks, hc=True, lcp=True, groups=1)
for i, b in enumerate(blocks):
    want = o.compress(b, hc=True)
    assert res[i] == len(want) and np.array_equal(dst[i, :res[i]], want), i
lib.emu_stats(st, 1)
print(f"dist {dist}, {nblk} blocks of 65536 bytes, bit-exact")
for i, nm in enumerate(NAMES):
    print(f"  {nm:55s} {st[i]:10d}  {st[i] / nblk:9.0f} per block  {st[i] / max(st[0], 1):6.3f} of the lane-steps")
