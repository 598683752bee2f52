"""The figures INTEGRATION.md section 4 quotes beside bench.py's: the reference's C on ONE host core (256 D2 blocks of 64 KiB, best of 5: fast encode,
LZ4HC, decode, time per block), the library's single-block calls (lz4hip_compress_limitedOutput / lz4hip_compressHC_limitedOutput /
lz4hip_uncompress: host pointers, one block, mean of 20 calls), and device-resident LZ4HC by batch size (default mapping, best of 2).
The oracle is used as bench.py's cpu_baseline leg uses it: as the thing compared with, never as the product.
usage: python tools/caller_expectations.py"""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lz4net_amd import batch, _lib
from oracle.oracle import Oracle, Reference

o = Oracle()
codec = Reference() if Reference.available() else o
n = 256
raw = o.gen(2, 77, 0, n)
lens = np.full(n, 65536, np.int32)
cap = 65536 + 65536 // 255 + 16
caps = np.full(n, cap, np.int32)
comp = np.zeros((n, cap + 16), np.uint8)
hcomp = np.zeros((n, cap + 16), np.uint8)
back = np.zeros((n, 65536 + 16), np.uint8)
best = {}
for _ in range(5):
    t, clen = o.batch(codec, "enc", raw, lens, comp, caps, threads=1); best["fast encode"] = min(best.get("fast encode", 9e9), t)
    t, hlen = o.batch(codec, "hc", raw, lens, hcomp, caps, threads=1); best["LZ4HC"] = min(best.get("LZ4HC", 9e9), t)
    t, _r = o.batch(codec, "dec", comp, clen, back, lens, threads=1); best["decode"] = min(best.get("decode", 9e9), t)
for k, t in best.items():
    print(f"one host core ({codec.kind} C), {n} D2 blocks of 64 KiB: {k:11s} {t / n * 1e6:8.1f} us per block = {n * 65536 / t / 1e9:6.3f} GB/s", flush=True)

L = _lib.lib()
src = np.ascontiguousarray(raw[3]); dst = np.zeros(cap, np.uint8); out = np.zeros(65536, np.uint8)
for name, fn, args in (("lz4hip_compress_limitedOutput", L.lz4hip_compress_limitedOutput, (src.ctypes.data, dst.ctypes.data, 65536, cap)),
                       ("lz4hip_compressHC_limitedOutput", L.lz4hip_compressHC_limitedOutput, (src.ctypes.data, dst.ctypes.data, 65536, cap))):
    fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    r = fn(*args)
    t0 = time.perf_counter()
    for _ in range(20): r = fn(*args)
    print(f"single-block call {name}: {(time.perf_counter() - t0) / 20 * 1e3:7.2f} ms per call (result {r})", flush=True)
L.lz4hip_compress_limitedOutput(src.ctypes.data, dst.ctypes.data, 65536, cap)
clen1 = L.lz4hip_compress_limitedOutput(src.ctypes.data, dst.ctypes.data, 65536, cap)
L.lz4hip_uncompress.restype = C.c_int; L.lz4hip_uncompress.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
r = L.lz4hip_uncompress(dst.ctypes.data, out.ctypes.data, 65536)
t0 = time.perf_counter()
for _ in range(20): r = L.lz4hip_uncompress(dst.ctypes.data, out.ctypes.data, 65536)
print(f"single-block call lz4hip_uncompress: {(time.perf_counter() - t0) / 20 * 1e3:7.2f} ms per call (result {r}, compressed {clen1}, bytes ok {bool(np.array_equal(out, src))})", flush=True)

nmax = 65536
rawd = batch.synth(2, 3, 0, nmax)
compd = torch.empty((nmax, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
for m in (256, 1024, 4096, 16384, 65536):
    batch.encode(rawd[:m], batch.BLOCK, compd[:m], batch.BOUND, hc=True); torch.cuda.synchronize()
    ts = []
    for _ in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); batch.encode(rawd[:m], batch.BLOCK, compd[:m], batch.BOUND, hc=True); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"LZ4HC device-resident, {m:6d} D2 blocks: {min(ts):8.1f} ms {m * 65536 / min(ts) / 1e6:7.2f} GB/s", flush=True)
