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


def test_header_is_plain_c(tmp_path):
    """include/lz4hip.h is the boundary a C / cgo / P-Invoke binding generator sees: it must compile as C99 on its
    own, and a C translation unit must link against the library using nothing but that header."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text(
        '#include "lz4hip.h"\n'
        "#include <stdio.h>\n"
        "int main(void) {\n"
        "    lz4hip_batch_t b = {0};\n"
        "    (void)b;\n"
        '    printf("%s %d %d\\n", lz4hip_codec_name(), lz4hip_compressBound(65536), lz4hip_device_count() >= 0);\n'
        "    return lz4hip_compressBound(65536) == 65809 ? 0 : 1;\n"
        "}\n")
    exe = tmp_path / "abi"
    lib_dir = os.path.join(root, "lz4net_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src),
                    "-L", lib_dir, "-llz4hip", "-Wl,-rpath," + lib_dir, "-o", str(exe)], check=True, timeout=120)
    r = subprocess.run([str(exe)], capture_output=True, timeout=60, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "65809" in r.stdout


def test_framing_host_logic_properties():
    """Header walks of the two container formats on synthetic headers (no codec call): varint round trips at the
    edges of every byte count, chunk lists of LZ4Stream frames, size-prefixed chunks of the legacy frame."""
    import random
    from lz4net_amd import stream as st, legacy_frame as lf
    rnd = random.Random(4)
    for _ in range(2000):
        v = rnd.choice([0, 1, 127, 128, 16383, 16384, (1 << 21) - 1, 1 << 21, (1 << 28) - 1, 1 << 28, (1 << 31) - 1]) + rnd.choice([0, 0, 1, -1]) * rnd.randint(0, 3)
        v = max(v, 0)
        enc = st.write_varint(v)
        assert st.read_varint(enc + b"\x00\x00", 0) == (v, len(enc)) and len(enc) == max(1, (v.bit_length() + 6) // 7)
    for _ in range(200):
        parts, frame, want = [], b"", []
        for _ in range(rnd.randint(0, 6)):
            n = rnd.randint(0, 40)
            payload = bytes(rnd.randrange(256) for _ in range(n))
            frame += st.write_varint(0) + st.write_varint(n)
            want.append((False, n, len(frame), n))
            frame += payload
            parts.append(payload)
        assert st.parse_chunks(frame) == want
        assert st.decompress_stream(frame) == b"".join(parts)              # raw chunks only: no codec involved
    for _ in range(200):
        frame, want = (lf.MAGIC).to_bytes(4, "little"), []
        for _ in range(rnd.randint(0, 6)):
            if rnd.random() < 0.2:
                frame += (lf.MAGIC).to_bytes(4, "little")                   # an appended frame's header
                continue
            n = rnd.randint(0, 50)
            frame += n.to_bytes(4, "little")
            want.append((len(frame), n))
            frame += bytes(rnd.randrange(256) for _ in range(n))
        assert lf.parse_frame(frame) == want


def test_tuning_knobs_without_gpu():
    """lz4hip_tuning_set / _get are host logic: named integer knobs, previous value returned, bad names and values refused.
    (The launch paths read these atomics and never call getenv(); the environment only seeds them, once.)"""
    from lz4net_amd import _lib
    L = _lib.lib()
    for name in ("decoder", "encoder", "hc"):
        prev = _lib.tuning_set(name, "lane")
        assert _lib.tuning_get(name) == 2
        assert _lib.tuning_set(name, "wave") == 2
        assert _lib.tuning_set(name, prev) == 1
        assert L.lz4hip_tuning_set(name.encode(), 3) == _lib.E_ARGUMENT
    for name in ("encoder_waves_per_cu", "hc_waves_per_cu", "hc_groups", "host_threads", "host_slices", "logical_devices"):
        prev = _lib.tuning_set(name, 5)
        assert _lib.tuning_get(name) == 5
        assert _lib.tuning_set(name, prev) == 5
        assert L.lz4hip_tuning_set(name.encode(), -1) == _lib.E_ARGUMENT
    assert L.lz4hip_tuning_set(b"no_such_knob", 1) == _lib.E_ARGUMENT and b"no_such_knob" in L.lz4hip_last_error()
    assert L.lz4hip_tuning_get(None) == _lib.E_ARGUMENT
    with _lib.tuning(decoder="lane", hc_groups=4):
        assert _lib.tuning_get("decoder") == 2 and _lib.tuning_get("hc_groups") == 4
    assert _lib.tuning_get("decoder") == 0 and _lib.tuning_get("hc_groups") == 0
    # read-only: what the lane encoder's table slab measured (no slab without a device: all zero, and not settable)
    for name in ("encoder_slab_rate", "encoder_slab_tried", "encoder_slab_chunks"):
        assert _lib.tuning_get(name) == 0
        assert L.lz4hip_tuning_set(name.encode(), 1) == _lib.E_ARGUMENT


def test_library_identifies_its_sources():
    """lz4hip_build_id() = the hash of lz4net_amd/csrc/ the binary was compiled from (prebuilt .so files travel to the GPU box): it equals the tree's
    hash after build(), the same string is found in the file without loading it (how build.is_stale() recognises a stale library by CONTENT),
    a library whose marker names other sources is stale, and bench.py's csrc_sha() is that hash."""
    import shutil
    import sys
    from lz4net_amd import _lib, build as hip_build
    L = _lib.lib()
    bid = L.lz4hip_build_id().decode()
    assert re.fullmatch(r"[0-9a-f]{16}(\+tuning)?", bid), bid
    assert bid.split("+")[0] == hip_build.csrc_sha()
    assert hip_build.built_id() == bid and not hip_build.is_stale()
    sys.path.insert(0, ROOT)
    import bench
    assert bench.csrc_sha() == hip_build.csrc_sha()
    # a copy whose marker names other sources: recognised as stale
    data = open(hip_build.SO, "rb").read()
    marker = b"LZ4HIP_BUILD_ID=" + bid.split("+")[0].encode()
    assert data.count(marker) >= 1
    so_dir = os.path.dirname(hip_build.SO)
    fake = os.path.join(so_dir, "liblz4hip.so.stale_test")
    try:
        with open(fake, "wb") as fh:
            fh.write(data.replace(marker, b"LZ4HIP_BUILD_ID=" + b"0123456789abcdef"))
        assert hip_build.built_id(fake) == "0123456789abcdef" + ("+tuning" if "+" in bid else "")
        assert hip_build.built_id(fake) != hip_build.wanted_id()
    finally:
        if os.path.exists(fake):
            os.remove(fake)
    assert hip_build.built_id(os.path.join(so_dir, "no_such_library.so")) is None


def test_decoder_store_knobs_without_gpu():
    """decoder_wrapped_stores is an ordinary settable knob; its read-only twin decoder_dual_store cannot be set and, without a device to probe,
    says 0 (the wrapped-row instantiation would run)."""
    from lz4net_amd import _lib
    L = _lib.lib()
    prev = _lib.tuning_set("decoder_wrapped_stores", 1)
    assert _lib.tuning_get("decoder_wrapped_stores") == 1
    assert _lib.tuning_set("decoder_wrapped_stores", prev) == 1
    assert L.lz4hip_tuning_set(b"decoder_dual_store", 1) == _lib.E_ARGUMENT
    if L.lz4hip_device_count() == 0:
        assert L.lz4hip_tuning_get(b"decoder_dual_store") == 0
