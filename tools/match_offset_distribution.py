"""Match offsets of the oracle's fast-encoder output per distribution: the share of matches whose source lies beyond a lane decoder's
output ring (ring bytes - 16), i.e. the far-match fetches a ring of that size leaves.  CPU only (TEST INFRASTRUCTURE: uses oracle/).
usage: python tools/match_offset_distribution.py [blocks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.oracle import Oracle

o = Oracle()
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rings = (64, 128, 192, 256, 512, 1024, 4096, 16384)
for dist in (2, 3):
    offs, lits = [], 0
    for i in range(nblk):
        blk = o.gen(dist, 1, i, 1)[0]
        c = np.asarray(o.compress(blk)).tolist(); ip = 0; L = len(c)
        while ip < L:
            t = c[ip]; ip += 1; ll = t >> 4
            if ll == 15:
                while True:
                    x = c[ip]; ip += 1; ll += x
                    if x != 255: break
            ip += ll; lits += ll
            if ip >= L: break
            offs.append(c[ip] | (c[ip + 1] << 8)); ip += 2
            if (t & 15) == 15:
                while True:
                    x = c[ip]; ip += 1
                    if x != 255: break
    a = np.array(offs)
    print(f"dist {dist}: {len(a) // nblk} matches per 64 KiB block, {lits / nblk:.0f} literal bytes; share of matches with offset > ring - 16:")
    print("   " + "  ".join(f"ring {r}: {float((a > r - 16).mean()):.3f}" for r in rings))
