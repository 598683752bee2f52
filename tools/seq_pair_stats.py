"""How often could the lane decoder retire TWO sequences in one iteration?  (CPU only, oracle = test infrastructure.)
Walks the sequences of fast-encoded 64 KiB blocks and counts, over non-overlapping greedy pairs (the way a two-sequence fast path would
take them), the pairs whose two headers lie inside the 16-byte view at the cursor (no length bytes), whose matches are both near
(offset <= ring - 20) and whose total output fits one append of N bytes.     usage: python tools/seq_pair_stats.py [dist] [blocks]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle


def sequences(c):
    """(literal length, match length, offset, header bytes incl. literals) per sequence of one LZ4 block; the final literal run is dropped"""
    out, ip, n = [], 0, len(c)
    while ip < n:
        p0 = ip
        tok = c[ip]; ip += 1
        ll = tok >> 4
        ext = False
        if ll == 15:
            ext = True
            while True:
                b = c[ip]; ip += 1; ll += b
                if b != 255: break
        ip += ll
        if ip >= n: break
        off = c[ip] | (c[ip + 1] << 8); ip += 2
        ml = tok & 15
        if ml == 15:
            ext = True
            while True:
                b = c[ip]; ip += 1; ml += b
                if b != 255: break
        out.append((ll, ml + 4, off, ip - p0, ext))
    return out


def main():
    dist = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    o = Oracle()
    near_max = 192 - 20
    tot = 0
    simple = 0
    pairs = {(v, cap): 0 for v in (16, 24) for cap in (16, 27, 32)}
    iters = {(v, cap): 0 for v in (16, 24) for cap in (16, 27, 32)}
    for i in range(nblk):
        seqs = sequences([int(x) for x in o.compress(o.gen(dist, 20260925, i, 1)[0])])
        tot += len(seqs)
        simple += sum(1 for (ll, ml, off, hb, ext) in seqs if not ext and ll <= 11 and ml <= 16 and 0 < off <= near_max)
        for (view, cap) in pairs:
            k = 0
            while k < len(seqs):
                a = seqs[k]
                b = seqs[k + 1] if k + 1 < len(seqs) else None
                ok = b is not None and not a[4] and not b[4] and a[3] + b[3] <= view and 0 < a[2] <= near_max and 0 < b[2] <= near_max \
                    and a[0] + a[1] + b[0] + b[1] <= cap and a[2] >= a[1] and b[2] >= b[1]          # (no self-overlapping copies in the fast path)
                if ok:
                    pairs[(view, cap)] += 1; k += 2
                else:
                    k += 1
                iters[(view, cap)] += 1
    print(f"dist {dist}: {nblk} blocks, {tot / nblk:.0f} sequences per block; {simple / tot:.3f} of them are 'simple' alone (header + <= 11 literals in the view, no length bytes, match <= 16 bytes, near)")
    for (view, cap) in sorted(pairs):
        print(f"  view {view:2d} bytes, one append of <= {cap:2d} bytes: {2 * pairs[(view, cap)] / tot:.3f} of the sequences retire in pairs "
              f"-> {iters[(view, cap)] / tot:.3f} iterations per sequence (one sequence per iteration = 1.000)")


main()
