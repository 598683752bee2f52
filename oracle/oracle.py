"""ctypes front-end for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import this module.
The product package (lz4net_amd/) never does: it fails loudly when the HIP library is missing.

``Oracle``    : the from-scratch C restatement (oracle/lz4_oracle.c)          -> kind "port"
``Reference`` : the reference's own C compiled in place (oracle/_ref/*.so)    -> kind "reference"
Both expose the same five calls so tests can run one against the other.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(HERE, "liblz4oracle.so")
_REF_SO = os.path.join(HERE, "_ref", "libref_lz4.so")

D_ZEROS, D_RANDOM, D_FUZ, D_RECORDS = 0, 1, 2, 3
DIST_NAMES = {D_ZEROS: "D0-zeros", D_RANDOM: "D1-incompressible", D_FUZ: "D2-fuzzer", D_RECORDS: "D3-records"}


def build(force: bool = False) -> None:
    """Compile liblz4oracle.so and (only when /root/reference is present) oracle/_ref."""
    srcs = [os.path.join(HERE, f) for f in ("lz4_oracle.c", "lz4_oracle.h", "synth.c", "synth.h", "batch.c", "Makefile")]
    stale = force or not os.path.exists(_ORACLE_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_ORACLE_SO) for s in srcs)
    need_ref = not os.path.exists(_REF_SO) and os.path.exists("/root/reference/original/lz4.c")
    if stale or need_ref:
        subprocess.run(["make", "-C", HERE, "-s"] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)


def _u8p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def compress_bound(n: int) -> int:
    return n + n // 255 + 16


class _Codec:
    """Common numpy-facing wrapper over a (compress, compressHC, uncompress, uncompress_unknown) set."""

    kind = "?"

    def __init__(self, lib, names):
        self.lib = lib
        self._enc = getattr(lib, names[0])
        self._hc = getattr(lib, names[1])
        self._dec = getattr(lib, names[2])
        self._decu = getattr(lib, names[3])
        for f in (self._enc, self._hc, self._decu):
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        self._dec.restype = C.c_int
        self._dec.argtypes = [C.c_void_p, C.c_void_p, C.c_int]

    # -- raw return-code API (what the parity tests compare) ---------------------------------
    def compress_raw(self, src: np.ndarray, cap: int, hc: bool = False, canary: int = 64):
        """Returns (ret, out_buffer_with_canary). The buffer is cap+canary bytes, canary = 0xA5."""
        src = np.ascontiguousarray(src, dtype=np.uint8)
        if hc:
            # The reference's HC sequence emitter tests its second output limit against the LITERAL
            # length (original/lz4hc.c:541), so with a too-small cap it can scribble up to
            # matchLength/255 bytes past `cap` before it finally returns 0.  Give it room.
            canary = max(canary, src.size // 255 + 64)
        out = np.full(max(cap, 0) + canary, 0xA5, dtype=np.uint8)
        pad = np.zeros(src.size + 16, dtype=np.uint8)   # the codec may peek a few bytes past short inputs
        pad[:src.size] = src
        ret = (self._hc if hc else self._enc)(pad.ctypes.data, out.ctypes.data, int(src.size), int(cap))
        return ret, out

    def compress(self, src: np.ndarray, hc: bool = False, cap: int | None = None) -> np.ndarray:
        src = np.ascontiguousarray(src, dtype=np.uint8)
        cap = compress_bound(src.size) if cap is None else cap
        ret, out = self.compress_raw(src, cap, hc)
        if ret <= 0:
            raise RuntimeError(f"compress returned {ret}")
        return out[:ret].copy()

    def uncompress_raw(self, comp: np.ndarray, osize: int, pad: int | None = None):
        comp = np.ascontiguousarray(comp, dtype=np.uint8)
        # known-size decode does not know its input size: a corrupt stream can run on through up to
        # 0.75*osize bytes of (zero) padding before the output fills up, so pad generously.
        pad = max(osize, 0) + 1024 if pad is None else pad
        buf = np.zeros(comp.size + pad, dtype=np.uint8)
        buf[:comp.size] = comp
        out = np.full(max(osize, 0) + 64, 0xA5, dtype=np.uint8)
        ret = self._dec(buf.ctypes.data, out.ctypes.data, int(osize))
        return ret, out

    def uncompress(self, comp: np.ndarray, osize: int) -> np.ndarray:
        ret, out = self.uncompress_raw(comp, osize)
        if ret != len(comp):
            raise RuntimeError(f"uncompress returned {ret}, expected {len(comp)}")
        return out[:osize].copy()

    def uncompress_unknown_raw(self, comp: np.ndarray, isize: int, max_out: int):
        comp = np.ascontiguousarray(comp, dtype=np.uint8)
        buf = np.zeros(max(comp.size, isize) + 64, dtype=np.uint8)
        buf[:comp.size] = comp
        out = np.full(max(max_out, 0) + 64, 0xA5, dtype=np.uint8)
        ret = self._decu(buf.ctypes.data, out.ctypes.data, int(isize), int(max_out))
        return ret, out

    def fn_ptr(self, which: str) -> int:
        f = {"enc": self._enc, "hc": self._hc, "dec": self._dec, "decu": self._decu}[which]
        return C.cast(f, C.c_void_p).value


class Oracle(_Codec):
    kind = "port"

    def __init__(self):
        build()
        lib = C.CDLL(_ORACLE_SO)
        super().__init__(lib, ("lz4o_compress_limited", "lz4o_compress_hc_limited",
                               "lz4o_uncompress", "lz4o_uncompress_unknown"))
        lib.lz4s_fill_batch.restype = None
        lib.lz4s_fill_batch.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_int64, C.c_void_p, C.c_int64, C.c_int]
        lib.lz4s_checksum.restype = C.c_uint64
        lib.lz4s_checksum.argtypes = [C.c_void_p, C.c_int64]
        lib.lz4o_batch_run.restype = C.c_double
        lib.lz4o_batch_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                       C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]

        lib.lz4o_verify_stream.restype = C.c_int64
        lib.lz4o_verify_stream.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_int, C.c_double]

    # -- full-corpus encoder check (nothing but lengths and checksums is kept) ----------------
    def verify_stream(self, codec: "_Codec", hc: bool, dist: int, seed: int, first: int, step: int, n: int,
                      length: int = 65536, threads: int = 1, budget_seconds: float = 0.0):
        """For synthetic blocks first + i*step, i < n: regenerate, compress with `codec`, checksum the compressed bytes.
        Returns (done, lens[int32], sums[uint64]); blocks [0, done) were processed (all of them unless the time budget ran out)."""
        lens = np.zeros(n, np.int32)
        sums = np.zeros(n, np.uint64)
        done = self.lib.lz4o_verify_stream(codec.fn_ptr("hc" if hc else "enc"), dist, seed, first, step, n, length,
                                           compress_bound(length), lens.ctypes.data, sums.ctypes.data, threads, budget_seconds)
        return int(done), lens, sums

    # -- synthetic data (CPU twin of the device generators) ----------------------------------
    def gen(self, dist: int, seed: int, first_block: int, n: int, length: int = 65536,
            stride: int | None = None) -> np.ndarray:
        stride = length if stride is None else stride
        out = np.zeros((n, stride), dtype=np.uint8)
        self.lib.lz4s_fill_batch(dist, seed, first_block, n, out.ctypes.data, stride, length)
        return out

    def checksum(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a, dtype=np.uint8)
        return int(self.lib.lz4s_checksum(a.ctypes.data, a.size))

    # -- threaded batch driver (cpu_baseline + fast sweeps) ----------------------------------
    def batch(self, codec: _Codec, which: str, src: np.ndarray, src_len: np.ndarray,
              dst: np.ndarray, dst_cap: np.ndarray, threads: int = 1):
        """Run `which` in {"enc","hc","dec","decu"} of `codec` over rows of src into rows of dst.
        Returns (elapsed_seconds, int32 results)."""
        n = src.shape[0]
        op = {"enc": 0, "hc": 0, "dec": 1, "decu": 2}[which]
        res = np.zeros(n, dtype=np.int32)
        src_len = np.ascontiguousarray(src_len, dtype=np.int32)
        dst_cap = np.ascontiguousarray(dst_cap, dtype=np.int32)
        t = self.lib.lz4o_batch_run(op, codec.fn_ptr(which), src.ctypes.data, src.strides[0],
                                    src_len.ctypes.data, dst.ctypes.data, dst.strides[0],
                                    dst_cap.ctypes.data, res.ctypes.data, n, threads)
        return t, res


class Reference(_Codec):
    kind = "reference"

    @staticmethod
    def available() -> bool:
        build()
        return os.path.exists(_REF_SO)

    def __init__(self):
        build()
        lib = C.CDLL(_REF_SO)
        super().__init__(lib, ("I64_LZ4_compress_limitedOutput", "I64_LZ4_compressHC_limitedOutput",
                               "I64_LZ4_uncompress", "I64_LZ4_uncompress_unknownOutputSize"))
