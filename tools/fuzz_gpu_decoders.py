"""Long differential fuzz of the decoders on the GPU (minutes; tests/test_gpu_parity.py::test_decoder_fuzz_slice runs a 30-second slice
of the same thing in every -m gpu run): tests/decoder_fuzz.py over many seeds.     usage: python tools/fuzz_gpu_decoders.py [seeds] [streams per seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import decoder_fuzz
from oracle.oracle import Oracle

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
per = int(sys.argv[2]) if len(sys.argv) > 2 else 400
o = Oracle()
total = bad = 0
t0 = time.time()
for seed in range(1000, 1000 + seeds):
    t, b = decoder_fuzz.run_seed(o, seed, per, report=lambda m: print(m, flush=True))
    total += t; bad += b
    print("seed %d done: %d comparisons so far, %d mismatches, %.0f s" % (seed, total, bad, time.time() - t0), flush=True)
print("TOTAL %d comparisons, %d mismatches" % (total, bad))
sys.exit(1 if bad else 0)
