"""ctypes binding of liblz4hip.so (the C ABI of include/lz4hip.h).

There is no fallback of any kind: if the shared library is missing it is built with hipcc, and if that
is impossible the import fails loudly.  Calls on a machine without a gfx950 device return
LZ4HIP_E_DEVICE, which the wrappers turn into Lz4HipError.
"""
from __future__ import annotations

import ctypes as C

from . import build as _build

E_DEVICE, E_ARGUMENT, E_MEMORY = -2000000001, -2000000002, -2000000003
MODE_FAST, MODE_HC = 0, 1


class Lz4HipError(RuntimeError):
    pass


class Batch(C.Structure):
    """struct lz4hip_batch (include/lz4hip.h)."""
    _fields_ = [("src", C.c_void_p), ("src_off", C.c_void_p), ("src_stride", C.c_int64), ("src_len", C.c_void_p),
                ("dst", C.c_void_p), ("dst_off", C.c_void_p), ("dst_stride", C.c_int64), ("dst_cap", C.c_void_p),
                ("dst_cap_all", C.c_int32), ("src_len_all", C.c_int32), ("result", C.c_void_p),
                ("n_blocks", C.c_int64)]


# every symbol include/lz4hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("lz4hip_codec_name", C.c_char_p, []),
    ("lz4hip_device_count", C.c_int, []),
    ("lz4hip_last_error", C.c_char_p, []),
    ("lz4hip_build_id", C.c_char_p, []),
    ("lz4hip_compressBound", C.c_int, [C.c_int]),
    ("lz4hip_dispatch_counts", C.c_int, [C.c_void_p, C.c_int]),
    ("lz4hip_release_workspaces", C.c_int, []),
    ("lz4hip_tuning_set", C.c_int, [C.c_char_p, C.c_int]),
    ("lz4hip_tuning_get", C.c_int, [C.c_char_p]),
    ("lz4hip_compress_limitedOutput", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    ("lz4hip_compress", C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    ("lz4hip_compressHC_limitedOutput", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    ("lz4hip_compressHC", C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    ("lz4hip_uncompress", C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    ("lz4hip_uncompress_unknownOutputSize", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    ("lz4hip_uncompress_bounded", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    ("lz4hip_encode_batch_device", C.c_int, [C.POINTER(Batch), C.c_int, C.c_void_p]),
    ("lz4hip_decode_batch_device", C.c_int, [C.POINTER(Batch), C.c_int, C.c_void_p]),
    ("lz4hip_encode_batch_host", C.c_int, [C.POINTER(Batch), C.c_int]),
    ("lz4hip_decode_batch_host", C.c_int, [C.POINTER(Batch), C.c_int]),
    ("lz4hip_encode_batch_host_multi", C.c_int, [C.POINTER(Batch), C.c_int, C.c_uint64]),
    ("lz4hip_decode_batch_host_multi", C.c_int, [C.POINTER(Batch), C.c_int, C.c_uint64]),
    ("lz4hip_synth_device", C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    ("lz4hip_checksum_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    ("lz4hip_compare_device", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        try:
            # torch wheels bundle their own libamdhip64 / libhsa-runtime64.  Loading torch FIRST makes the
            # dynamic linker resolve our DT_NEEDED entries to those same copies, so the process has ONE
            # HIP/HSA runtime (two HSA runtimes in one process do not both see the GPU).
            import torch  # noqa: F401
        except ImportError:
            pass
        so = _build.build()                     # raises if the library cannot be produced
        handle = C.CDLL(so)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(handle, name)          # AttributeError == ABI drift: fail loudly
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


K_DECODE_WAVE, K_DECODE_LANE, K_ENCODE_WAVE, K_ENCODE_LANE, K_HC_WAVE, K_HC_LANE, K_COUNT = range(7)


def dispatch_counts() -> list:
    """Launches per kernel family since the library was loaded (lz4hip_dispatch_counts)."""
    buf = (C.c_uint64 * K_COUNT)()
    n = lib().lz4hip_dispatch_counts(buf, K_COUNT)
    assert n == K_COUNT
    return list(buf)


_MAPPING = {"auto": 0, "wave": 1, "lane": 2}


def tuning_set(name: str, value) -> int:
    """lz4hip_tuning_set; mapping knobs also take "auto" / "wave" / "lane".  Returns the previous value."""
    v = _MAPPING[value] if isinstance(value, str) else int(value)
    return check(lib().lz4hip_tuning_set(name.encode(), v))


def tuning_get(name: str) -> int:
    return check(lib().lz4hip_tuning_get(name.encode()))


class tuning:
    """with tuning(decoder="lane", hc_groups=4): ...  -- sets knobs, restores the previous values on exit."""

    def __init__(self, **knobs):
        self.knobs, self.prev = knobs, {}

    def __enter__(self):
        for k, v in self.knobs.items():
            self.prev[k] = tuning_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            tuning_set(k, v)
        return False


def check(rc: int) -> int:
    """Raise on library-level failures (LZ4HIP_E_*); pass codec results through."""
    if rc <= E_DEVICE and rc >= E_MEMORY:
        raise Lz4HipError(f"liblz4hip error {rc}: {lib().lz4hip_last_error().decode(errors='replace')}")
    return rc
