"""CPU-only: liblz4hip.so builds for gfx950 (hipcc cross-compiles), loads, and exports every symbol that
include/lz4hip.h declares; without a device the codec entry points fail LOUDLY (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    from lz4net_amd import _lib
    L = _lib.lib()
    header = open(os.path.join(ROOT, "include", "lz4hip.h")).read()
    declared = set(re.findall(r"\b(lz4hip_[A-Za-z0-9_]+)\s*\(", header))
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(L, name), name


def test_host_logic_without_gpu():
    from lz4net_amd import LZ4Codec, _lib
    from lz4net_amd.codec import ArgumentException, ArgumentNullException
    import pytest
    L = _lib.lib()
    assert L.lz4hip_compressBound(65536) == 65809 and LZ4Codec.MaximumOutputLength(65536) == 65809
    # argument checking is host logic and must behave like CheckArguments (src/LZ4ps/LZ4Codec.cs:151-170)
    assert LZ4Codec.Encode(b"", 0, 0, bytearray(4), 0, 4) == 0
    with pytest.raises(ArgumentNullException):
        LZ4Codec.Encode(None, 0, 3, bytearray(4), 0, 4)
    with pytest.raises(ArgumentException):
        LZ4Codec.Encode(b"abc", 1, 3, bytearray(4), 0, 4)
    with pytest.raises(ArgumentException):
        LZ4Codec.Decode(b"abc", 0, 3, bytearray(4), 2, 4, True)
    with pytest.raises(ArgumentException):
        LZ4Codec.Unwrap(b"1234567")
    if L.lz4hip_device_count() == 0:
        a = np.zeros(100, np.uint8)
        o = np.zeros(200, np.uint8)
        rc = L.lz4hip_compress_limitedOutput(a.ctypes.data, o.ctypes.data, 100, 200)
        assert rc == _lib.E_DEVICE and b"device" in L.lz4hip_last_error().lower()
        with pytest.raises(_lib.Lz4HipError):
            LZ4Codec.Encode(b"hello hello hello hello", 0, 23)


def test_product_never_imports_the_oracle():
    # the oracle is test infrastructure: nothing under lz4net_amd/ or include/ may reference it
    bad = []
    for base in ("lz4net_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hpp", ".hip", ".h")):
                    text = open(os.path.join(d, f), errors="replace").read()
                    if re.search(r"^\s*(import|from)\s+oracle|#include\s*[<\"][^>\"]*(simt|oracle)|lz4o_\w+\s*\(|liblz4oracle|libref_lz4",
                                 text, re.M):
                        bad.append(os.path.join(d, f))
    assert not bad, bad
