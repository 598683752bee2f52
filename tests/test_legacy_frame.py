"""Legacy LZ4 CLI frame (original/lz4demo.c:84-87,166-317) and batched Wrap/Unwrap (src/LZ4/LZ4Codec.cs:471-599).
CPU: header walk against frames written by the reference's own CLI (oracle/_ref/lz4demo).  GPU: frames written by
the product are byte-identical to the CLI's and decode with it; frames written by the CLI decode with the product."""
import os
import subprocess

import numpy as np
import pytest

from lz4net_amd import legacy_frame as lf

CLI = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "lz4demo")


def _cli(tmp_path, args, data):
    src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
    src.write_bytes(data)
    if dst.exists():
        dst.unlink()
    r = subprocess.run([CLI, *args, str(src), str(dst)], capture_output=True, timeout=120)
    return r.returncode, (dst.read_bytes() if dst.exists() else b"")


needs_cli = pytest.mark.skipif(not os.path.exists(CLI), reason="reference CLI not built (oracle/_ref/lz4demo)")


def _sample(oracle, n_bytes, dist=3):
    blocks = (n_bytes + 65535) // 65536
    return oracle.gen(dist, 11, 0, blocks).reshape(-1)[:n_bytes].tobytes()


@needs_cli
def test_parse_cli_frames(oracle, tmp_path):
    data = _sample(oracle, 300000)
    rc, frame = _cli(tmp_path, ["-c0"], data)
    assert rc == 0 and frame[:4] == bytes([0x02, 0x21, 0x4C, 0x18])
    chunks = lf.parse_frame(frame)
    assert len(chunks) == 1 and chunks[0] == (8, len(frame) - 8)
    assert int.from_bytes(frame[4:8], "little") == len(frame) - 8
    # the payload is exactly LZ4_compress of the whole input (one chunk < 8 MiB)
    ret, out = oracle.compress_raw(np.frombuffer(data, np.uint8), lf._bound(len(data)))
    assert ret == len(frame) - 8 and bytes(out[:ret]) == frame[8:]
    # appended frames: the second magic is skipped like lz4demo.c:289-290 does
    both = lf.parse_frame(frame + frame)
    assert both == [(8, len(frame) - 8), (len(frame) + 8, len(frame) - 8)]
    assert lf.parse_frame(frame[:4]) == []
    with pytest.raises(lf.ArgumentException):
        lf.parse_frame(b"\x00\x00\x00\x00" + frame[4:])
    with pytest.raises(lf.ArgumentException):
        lf.parse_frame(frame[:-1])
    with pytest.raises(lf.ArgumentException):
        lf.parse_frame(frame + b"\x01\x02")


def test_bound_matches_reference_macro():
    for n in (0, 1, 254, 255, 256, 65536, 8 << 20):
        assert lf._bound(n) == n + n // 255 + 16                   # original/lz4.h:85-86


@pytest.mark.gpu
@needs_cli
@pytest.mark.parametrize("hc", [False, True])
def test_frame_interop_with_reference_cli(oracle, tmp_path, hc):
    for n_bytes, dist in ((1000, 2), (70000, 2), (300000, 3), ((8 << 20) + 123457, 3)):
        if hc and n_bytes > (1 << 20):
            continue                                                # HC over 8 MiB chunks: covered by the fast path only (time)
        data = _sample(oracle, n_bytes, dist)
        ours = lf.compress_frame(data, high_compression=hc)
        rc, theirs = _cli(tmp_path, ["-c1" if hc else "-c0"], data)
        assert rc == 0
        assert ours == theirs, "frame differs from the reference CLI's at %d bytes" % n_bytes
        rc, back = _cli(tmp_path, ["-d"], ours)                     # their reader on our frame
        assert rc == 0 and back == data
        assert lf.decompress_frame(theirs) == data                  # our reader on their frame
        assert lf.decompress_frame(ours + ours) == data + data      # appended frames


@pytest.mark.gpu
def test_frame_small_chunks_roundtrip(oracle):
    data = _sample(oracle, 500000, 2)
    for chunk in (4096, 65536, 100000):
        frame = lf.compress_frame(data, chunk_size=chunk)
        assert len(lf.parse_frame(frame)) == (len(data) + chunk - 1) // chunk
        assert lf.decompress_frame(frame, chunk_size=chunk) == data
    assert lf.compress_frame(b"") == (0x184C2102).to_bytes(4, "little")
    assert lf.decompress_frame(lf.compress_frame(b"")) == b""
    with pytest.raises(lf.ArgumentException):                       # corrupted payload -> negative decoder result
        bad = bytearray(lf.compress_frame(data[:70000]))
        bad[8] = 0xFF; bad[9] = 0xFF; bad[10] = 0xFF
        lf.decompress_frame(bytes(bad))


@pytest.mark.gpu
@pytest.mark.parametrize("hc", [False, True])
def test_wrap_many_equals_wrap(oracle, hc):
    from lz4net_amd.codec import LZ4Codec
    rng = np.random.default_rng(5)
    msgs = [b"", b"a", bytes(100), rng.integers(0, 256, 5000, dtype=np.uint8).tobytes(),
            _sample(oracle, 20000, 2), _sample(oracle, 65536, 3), b"Lorem ipsum dolor sit amet, " * 40]
    many = LZ4Codec.WrapMany(msgs, high_compression=hc)
    single = [(LZ4Codec.WrapHC if hc else LZ4Codec.Wrap)(m) for m in msgs]
    assert many == single
    assert LZ4Codec.UnwrapMany(many) == msgs
    assert [LZ4Codec.Unwrap(w) for w in many] == msgs
