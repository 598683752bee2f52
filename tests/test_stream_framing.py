"""LZ4Stream chunk framing (src/LZ4/LZ4Stream.cs:162-312) over the batch API.
CPU: varints, header walk, raw-only streams (no codec call).  GPU: round trips, payloads bit-exact to the oracle."""
import numpy as np
import pytest

from lz4net_amd import stream as st


def test_varints():
    for v in (0, 1, 127, 128, 300, 16383, 16384, 65536, 1 << 20, (1 << 31) - 1):
        enc = st.write_varint(v)
        assert st.read_varint(enc, 0) == (v, len(enc))
    assert st.write_varint(300) == bytes([0xAC, 0x02])
    assert st.read_varint(b"", 0) == (None, 0)                 # clean end of stream
    with pytest.raises(st.EndOfStreamException):
        st.read_varint(bytes([0x80]), 0)                        # truncated inside a varint


def test_raw_only_stream_needs_no_codec():
    a, b = b"0123456789abcdef", b"xyz"
    framed = st.write_varint(0) + st.write_varint(len(a)) + a + st.write_varint(0) + st.write_varint(len(b)) + b
    assert st.parse_chunks(framed) == [(False, 16, 2, 16), (False, 3, 20, 3)]
    assert st.decompress_stream(framed) == a + b
    with pytest.raises(st.EndOfStreamException):
        st.parse_chunks(framed[:-1])                            # payload shorter than its header says
    with pytest.raises(st.EndOfStreamException):                # compressedLength > originalLength == corrupted
        st.parse_chunks(st.write_varint(1) + st.write_varint(4) + st.write_varint(9) + b"123456789")


@pytest.mark.gpu
@pytest.mark.parametrize("hc", [False, True])
def test_stream_roundtrip_and_payload_parity(oracle, hc):
    rng = np.random.default_rng(3)
    data = np.concatenate([oracle.gen(2, 1, 0, 3)[:, :].reshape(-1), rng.integers(0, 256, 70000, dtype=np.uint8),
                           np.zeros(100000, np.uint8), oracle.gen(3, 2, 5, 2).reshape(-1)]).tobytes()
    for block_size in (16, 4096, 65536, 1 << 20):
        if block_size == 16:
            payload = data[:4000]
        else:
            payload = data
        framed = st.compress_stream(payload, block_size, high_compression=hc)
        assert st.decompress_stream(framed) == payload
        # every compressed chunk carries exactly the bytes the reference encoder produces with outputLength = inputLength
        pos = 0
        for compressed, original, off, ln in st.parse_chunks(framed):
            chunk = np.frombuffer(payload[pos:pos + original], dtype=np.uint8)
            ret, out = oracle.compress_raw(chunk, original, hc=hc)
            if compressed:
                assert ret == ln and bytes(out[:ret]) == framed[off:off + ln]
            else:
                assert ret <= 0 or ret >= original
                assert framed[off:off + ln] == payload[pos:pos + original]
            pos += original
        assert pos == len(payload)
