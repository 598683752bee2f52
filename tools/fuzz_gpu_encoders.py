"""Differential fuzz of the fast encoders on the GPU against the CPU oracle: blocks that stress the exactness arguments of the wavefront-mapped
encoder's second version (register window, combined test + count, merged catch-up, one-store emit) and of the lane-mapped one -- tiny alphabets,
runs of period 1-5 between junk, copies of earlier content, fuzzer-style and record-like rows, long runs with single disturbed bytes, sizes from 13
bytes to just above LZ4_64KLIMIT -- with full and with too-small output limits (return value, bytes, guard bytes).  Through the host-pointer C ABI.
usage: python tools/fuzz_gpu_encoders.py [rounds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import gpu_helpers as gpu
from lz4net_amd import _lib
from oracle.oracle import Oracle

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 606
o = Oracle()
rng = np.random.default_rng(seed)
total = bad = 0
t0 = time.time()
for r in range(rounds):
    blocks = []
    for i in range(192):
        mode = int(rng.integers(0, 7))
        sz = int(rng.integers(13, 6000)) if rng.integers(0, 8) else int(rng.integers(30000, 65560))
        if mode == 0:
            row = rng.integers(0, int(rng.integers(2, 4)), sz).astype(np.uint8)
        elif mode == 1:
            row = rng.integers(0, 256, sz).astype(np.uint8); pos = 0
            while pos < sz:
                per = int(rng.integers(1, 6)); ln = int(rng.integers(4, 700))
                pat = rng.integers(0, 3, per).astype(np.uint8)
                seg = np.tile(pat, ln // per + 2)[:ln]; e = min(sz, pos + ln); row[pos:e] = seg[:e - pos]; pos = e + int(rng.integers(0, 12))
        elif mode == 2:
            row = rng.integers(0, 8, sz).astype(np.uint8); pos = 64
            while pos < sz - 8:
                ln = int(rng.integers(4, 400)); src = int(rng.integers(0, pos)); e = min(sz, pos + ln)
                for j in range(pos, e):
                    row[j] = row[src + (j - pos)] if src + (j - pos) < j else row[j]
                pos = e + int(rng.integers(0, 6))
        elif mode == 3:
            row = o.gen(2, seed * 131 + r, i, 1, max(sz, 16))[0][:sz].copy()
        elif mode == 4:
            row = o.gen(3, seed * 131 + r, i, 1, max(sz, 16))[0][:sz].copy()
            if rng.integers(0, 2):
                row[:sz // 3] = row[0]
        elif mode == 5:
            row = np.full(sz, int(rng.integers(0, 256)), np.uint8)
            for _ in range(int(rng.integers(0, 40))):
                row[int(rng.integers(0, sz))] = int(rng.integers(0, 256))
        else:
            row = rng.integers(0, 256, sz).astype(np.uint8)
            for _ in range(sz // 150):
                src, ln, dstp = int(rng.integers(0, max(sz - 40, 1))), int(rng.integers(4, 40)), int(rng.integers(0, max(sz - 40, 1)))
                if dstp > src and dstp + ln <= sz:
                    row[dstp:dstp + ln] = row[src:src + ln]
        blocks.append(row)
    want = [o.compress(a) for a in blocks]
    for mapping, wg5 in (("wave", 1), ("wave", 2), ("lane", 0)):      # one block per workgroup, five per workgroup (launch_encode's form where that saves a residency round), lane per block
        with _lib.tuning(encoder=mapping, encoder_wg5=wg5):
            for delta in (None, 0, -1, -5):
                caps = None if delta is None else [max(len(w) + delta, 0) for w in want]
                res, dst = gpu.encode(blocks, caps=caps)
                for i, a in enumerate(blocks):
                    total += 1
                    cap = (a.size + a.size // 255 + 16) if caps is None else caps[i]
                    exp = len(want[i]) if caps is None else o.compress_raw(a, cap)[0]
                    ok = res[i] == exp and (exp <= 0 or np.array_equal(dst[i, :exp], want[i])) and (dst[i, cap:] == 0xA5).all()
                    if not ok:
                        bad += 1
                        if bad < 8:
                            print("MISMATCH seed", seed, "round", r, "block", i, "size", a.size, mapping, wg5, delta, res[i], exp, flush=True)
                            np.save(f"/tmp/enc_fuzz_bad_{seed}_{r}_{i}.npy", a)
    print("round %d done: %d comparisons so far, %d mismatches, %.0f s" % (r, total, bad, time.time() - t0), flush=True)
print("TOTAL %d comparisons, %d mismatches" % (total, bad))
sys.exit(1 if bad else 0)
