"""Long differential fuzz of the decoders on the GPU (not a pytest test: minutes, not seconds): arbitrary LZ4 streams from tests/stream_fuzz.py -- well
formed, truncated, extended, corrupted, with offset 0 -- through the host-pointer C ABI with the wavefront mapping (bursts included), the lane mapping
(one block per lane) and the persistent lane grid with ONE wavefront (every lane restarts many times); known and unknown output size; results and bytes
against the CPU oracle, canaries behind every row.     usage: python tools/fuzz_gpu_decoders.py [seeds] [streams per seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import gpu_helpers as gpu
import stream_fuzz
from lz4net_amd import _lib
from oracle.oracle import Oracle

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
per = int(sys.argv[2]) if len(sys.argv) > 2 else 400
o = Oracle()
total = bad = 0
t0 = time.time()
for seed in range(1000, 1000 + seeds):
    cs = stream_fuzz.cases(seed, per) + stream_fuzz.cases(seed + 5000, per // 8, max_size=30000)
    comps = [c for (c, _), _ in cs]
    sizes = [t for _, t in cs]
    holes = [stream_fuzz.has_zero_offset(c, t + 8) for c, t in zip(comps, sizes)]
    want_k = [o.uncompress_raw(c, t) for c, t in zip(comps, sizes)]
    caps = [t + (i % 3) * 7 - (5 if i % 11 == 0 else 0) for i, t in enumerate(sizes)]
    want_u = [o.uncompress_unknown_raw(c, len(c), cap) for c, cap in zip(comps, caps)]
    pad = [np.concatenate([c, np.zeros(t + 1024, np.uint8)]) for c, t in zip(comps, sizes)]
    padu = [np.concatenate([c, np.zeros(8, np.uint8)]) for c in comps]
    for name, knobs in (("wave", dict(decoder="wave")), ("lane", dict(decoder="lane", decoder_persist=2)),
                        ("persistent x1", dict(decoder="lane", decoder_persist=1, decoder_groups=1))):
        with _lib.tuning(**knobs):
            res, dst = gpu.decode(pad, sizes, known=True)
            for i, (w, out) in enumerate(want_k):
                ok = res[i] == w and (dst[i, sizes[i]:] == 0xA5).all() and (w < 0 or holes[i] or np.array_equal(dst[i, :sizes[i]], out[:sizes[i]]))
                total += 1
                if not ok:
                    bad += 1
                    print("MISMATCH known", name, "seed", seed, "case", i, "result", res[i], "want", w, flush=True)
            res, dst = gpu.decode(padu, caps, known=False, src_lens=[len(c) for c in comps])
            for i, (w, out) in enumerate(want_u):
                ok = res[i] == w and (dst[i, max(caps[i], 0):] == 0xA5).all() and (w < 0 or holes[i] or np.array_equal(dst[i, :w], out[:w]))
                total += 1
                if not ok:
                    bad += 1
                    print("MISMATCH unknown", name, "seed", seed, "case", i, "result", res[i], "want", w, flush=True)
    print("seed %d done: %d comparisons so far, %d mismatches, %.0f s" % (seed, total, bad, time.time() - t0), flush=True)
print("TOTAL %d comparisons, %d mismatches" % (total, bad))
sys.exit(1 if bad else 0)
