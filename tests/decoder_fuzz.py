"""Differential fuzz of the decoder forms on the GPU (shared by tests/test_gpu_parity.py::test_decoder_fuzz_slice and
tools/fuzz_gpu_decoders.py): arbitrary LZ4 streams from tests/stream_fuzz.py -- well formed, truncated, extended, corrupted, with
offset 0; offsets around every ring / window / burst threshold -- through the host-pointer C ABI with the wavefront mapping (bursts
included), the lane mapping (one block per lane: workgroups of one wavefront, and -- round 6 -- of four wavefronts with wrapped ring rows) and the
persistent lane grid with ONE wavefront (every lane restarts many times);
known and unknown output size; results and bytes against the CPU oracle, canaries behind every row."""
import time

import numpy as np

import gpu_helpers as gpu
import stream_fuzz

FORMS = (("wave", dict(decoder="wave")), ("lane", dict(decoder="lane", decoder_persist=2, decoder_wg4=1)),
         ("lane, four wavefronts per workgroup", dict(decoder="lane", decoder_persist=2, decoder_wg4=2)),
         ("persistent x1", dict(decoder="lane", decoder_persist=1, decoder_groups=1)))


def run_seed(o, seed, per, report=print):
    """One seed's streams through every decoder form; returns (comparisons, mismatches)."""
    from lz4net_amd import _lib
    total = bad = 0
    cs = stream_fuzz.cases(seed, per) + stream_fuzz.cases(seed + 5000, per // 8, max_size=30000)
    comps = [c for (c, _), _ in cs]
    sizes = [t for _, t in cs]
    holes = [stream_fuzz.has_zero_offset(c, t + 8) for c, t in zip(comps, sizes)]
    want_k = [o.uncompress_raw(c, t) for c, t in zip(comps, sizes)]
    caps = [t + (i % 3) * 7 - (5 if i % 11 == 0 else 0) for i, t in enumerate(sizes)]
    want_u = [o.uncompress_unknown_raw(c, len(c), cap) for c, cap in zip(comps, caps)]
    pad = [np.concatenate([c, np.zeros(t + 1024, np.uint8)]) for c, t in zip(comps, sizes)]
    padu = [np.concatenate([c, np.zeros(8, np.uint8)]) for c in comps]
    for name, knobs in FORMS:
        with _lib.tuning(**knobs):
            res, dst = gpu.decode(pad, sizes, known=True)
            for i, (w, out) in enumerate(want_k):
                ok = res[i] == w and (dst[i, sizes[i]:] == 0xA5).all() and (w < 0 or holes[i] or np.array_equal(dst[i, :sizes[i]], out[:sizes[i]]))
                total += 1
                if not ok:
                    bad += 1
                    report(f"MISMATCH known {name} seed {seed} case {i} result {res[i]} want {w}")
            res, dst = gpu.decode(padu, caps, known=False, src_lens=[len(c) for c in comps])
            for i, (w, out) in enumerate(want_u):
                ok = res[i] == w and (dst[i, max(caps[i], 0):] == 0xA5).all() and (w < 0 or holes[i] or np.array_equal(dst[i, :w], out[:w]))
                total += 1
                if not ok:
                    bad += 1
                    report(f"MISMATCH unknown {name} seed {seed} case {i} result {res[i]} want {w}")
    return total, bad


def run(o, first_seed, seeds, per, seconds=None, report=print):
    """Seeds first_seed .. first_seed + seeds - 1 (stops early after `seconds`, but never before three seeds)."""
    total = bad = done = 0
    t0 = time.time()
    for seed in range(first_seed, first_seed + seeds):
        t, b = run_seed(o, seed, per, report)
        total += t; bad += b; done += 1
        if seconds is not None and done >= 3 and time.time() - t0 > seconds:
            break
    return total, bad, done
