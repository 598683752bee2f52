"""Lane-iteration accounting of the fourth-generation lane decoder under the SIMT emulator (TEST INFRASTRUCTURE): how many
iterations a wavefront needs for 64 blocks and what its lanes do in them.  usage: python tools/emu_decoder_stats.py [dist] [cfg,cfg,...] [block bytes]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu_helpers as emu
from oracle.oracle import Oracle

NAMES = ["lane-iterations", "lane not done", "  input not staged", "flush needed", "  flush needed, no slot", "copy in progress", "  copy: nothing appended",
         "  copy: ring full (no room)", "  copy: far chunk not fetched", "  copy continues next iteration", "parsed a sequence", "holds a parsed sequence", "promoted a sequence",
         "  promote blocked: ring full", "idle: nothing to copy, nothing parsed", "far fetch issued", "far fetch wanted, source not flushed yet"]
dist = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfgs = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "3192,7192").split(",")]
size = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
o = Oracle()
blocks = [o.gen(dist, 99, i, 1, size)[0] for i in range(64)]
comps = [o.compress(b) for b in blocks]
nseq = None
for cfg in cfgs:
    lib = emu.lib()
    st = (C.c_ulonglong * 32)()
    lib.emu_stats(st, 1)
    lib.emu_iterations.restype = C.c_ulonglong
    lib.emu_iterations(1)
    res, dst = emu.decode([np.concatenate([c, np.zeros(64, np.uint8)]) for c in comps], [b.size for b in blocks], known=True, lane=cfg, gen=4)
    assert all(res[i] == len(comps[i]) and np.array_equal(dst[i, :size], blocks[i]) for i in range(64))
    it = lib.emu_iterations(1)
    lib.emu_stats(st, 1)
    print(f"dist {dist}, 64 blocks of {size} bytes, configuration {cfg}: {it} wave-iterations")
    for i, nm in enumerate(NAMES):
        print(f"  {nm:45s} {st[i]:10d}  {st[i] / max(st[0], 1):6.3f} of the lane-iterations, {st[i] / 64 / it * 100:5.1f} % of lanes per iteration")
