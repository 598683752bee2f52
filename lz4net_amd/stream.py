"""Batched LZ4Stream chunk framing -- the natural producer/consumer of block batches.

Mirrors the wire format of lz4net's ``LZ4Stream`` (src/LZ4/LZ4Stream.cs): a stream is a sequence of chunks

    varint(flags)  varint(originalLength)  [varint(compressedLength) if flags & Compressed]  payload

with ``ChunkFlags { None = 0, Compressed = 1, HighCompression = 2 }`` (src/LZ4/LZ4Stream.cs:43-59), varints
as in WriteVarInt / TryReadVarInt (:162-218), a chunk stored raw when the encoder returns <= 0 or does not
shrink it -- the encoder is called with ``outputLength = inputLength`` (FlushCurrentChunk, :239-269) -- and
``Decode(..., knownOutputLength: true)`` on the way back (AcquireNextChunk, :274-312).

The reference encodes and decodes one chunk per call; here all chunks of a buffer go through ONE
lz4hip_encode_batch_host / lz4hip_decode_batch_host call (include/lz4hip.h), i.e. one batch on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .codec import ArgumentException

FLAG_COMPRESSED, FLAG_HIGH_COMPRESSION = 1, 2
DEFAULT_BLOCK_SIZE = 1024 * 1024          # LZ4Stream's default (src/LZ4/LZ4Stream.cs:127-139); minimum 16


class EndOfStreamException(ArgumentException):
    """LZ4Stream.EndOfStream(): the stream is truncated or corrupted."""


def write_varint(value: int) -> bytes:
    out = bytearray()
    while True:
        b = value & 0x7F
        value >>= 7
        out.append(b | (0x80 if value else 0))
        if not value:
            return bytes(out)


def read_varint(buf, pos: int):
    """Returns (value, new_pos) or (None, pos) at a clean end of stream (TryReadVarInt)."""
    result, count, start = 0, 0, pos
    while True:
        if pos >= len(buf):
            if count == 0:
                return None, start
            raise EndOfStreamException("unexpected end of stream inside a varint")
        b = buf[pos]
        pos += 1
        result += (b & 0x7F) << count
        count += 7
        if (b & 0x80) == 0 or count >= 64:
            return result, pos


def _batch(src, src_off, src_len, dst, dst_off, dst_cap, result):
    n = len(src_len)
    return _lib.Batch(src=src.ctypes.data, src_off=src_off.ctypes.data, src_stride=0, src_len=src_len.ctypes.data,
                      dst=dst.ctypes.data, dst_off=dst_off.ctypes.data, dst_stride=0, dst_cap=dst_cap.ctypes.data,
                      dst_cap_all=0, src_len_all=0, result=result.ctypes.data, n_blocks=n)


def compress_stream(data, block_size: int = DEFAULT_BLOCK_SIZE, high_compression: bool = False) -> bytes:
    """LZ4Stream(Compress).Write(data) + Close(): every chunk of `data` encoded in one GPU batch."""
    block_size = max(16, int(block_size))
    raw = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    n = (raw.size + block_size - 1) // block_size
    if n == 0:
        return b""
    offs = np.arange(n, dtype=np.int64) * block_size
    lens = np.minimum(block_size, raw.size - offs).astype(np.int32)
    comp = np.zeros(raw.size, dtype=np.uint8)                  # outputLength = inputLength per chunk, packed at the same offsets
    res = np.zeros(n, dtype=np.int32)
    b = _batch(raw, offs, lens, comp, offs, lens, res)
    _lib.check(_lib.lib().lz4hip_encode_batch_host(C.byref(b), _lib.MODE_HC if high_compression else _lib.MODE_FAST))
    out = bytearray()
    for i in range(n):
        o, ln, cl = int(offs[i]), int(lens[i]), int(res[i])
        compressed = 0 < cl < ln
        flags = (FLAG_COMPRESSED if compressed else 0) | (FLAG_HIGH_COMPRESSION if high_compression else 0)
        out += write_varint(flags) + write_varint(ln)
        if compressed:
            out += write_varint(cl)
            out += comp[o:o + cl].tobytes()
        else:
            out += raw[o:o + ln].tobytes()
    return bytes(out)


def parse_chunks(stream):
    """Header walk of a framed stream -> list of (is_compressed, original_length, payload_offset, payload_length)."""
    buf = memoryview(stream) if not isinstance(stream, memoryview) else stream
    pos, chunks = 0, []
    while True:
        flags, pos = read_varint(buf, pos)
        if flags is None:
            return chunks
        original, pos = read_varint(buf, pos)
        if original is None:
            raise EndOfStreamException("missing chunk length")
        compressed = bool(flags & FLAG_COMPRESSED)
        if compressed:
            clen, pos = read_varint(buf, pos)
            if clen is None:
                raise EndOfStreamException("missing compressed length")
        else:
            clen = original
        original, clen = original & 0xFFFFFFFF, clen & 0xFFFFFFFF
        if original >= 1 << 31:
            original -= 1 << 32
        if clen >= 1 << 31:
            clen -= 1 << 32
        if clen > original or clen < 0:
            raise EndOfStreamException("corrupted chunk header")
        if pos + clen > len(buf):
            raise EndOfStreamException("truncated chunk payload")
        if compressed and (flags >> 2) != 0:
            raise NotImplementedError("Chunks with multiple passes are not supported.")
        chunks.append((compressed, original, pos, clen))
        pos += clen


def decompress_stream(stream) -> bytes:
    """LZ4Stream(Decompress).Read to end: all compressed chunks decoded in one GPU batch."""
    data = np.frombuffer(bytes(stream), dtype=np.uint8)
    chunks = parse_chunks(memoryview(bytes(stream)))
    total = sum(c[1] for c in chunks)
    out = np.zeros(total, dtype=np.uint8)
    out_off, pos = [], 0
    for _, original, _, _ in chunks:
        out_off.append(pos)
        pos += original
    idx = [i for i, c in enumerate(chunks) if c[0]]
    for i, (compressed, original, off, ln) in enumerate(chunks):
        if not compressed:
            out[out_off[i]:out_off[i] + original] = data[off:off + ln]
    if idx:
        src_off = np.array([chunks[i][2] for i in idx], dtype=np.int64)
        src_len = np.array([chunks[i][3] for i in idx], dtype=np.int32)
        dst_off = np.array([out_off[i] for i in idx], dtype=np.int64)
        dst_len = np.array([chunks[i][1] for i in idx], dtype=np.int32)
        res = np.zeros(len(idx), dtype=np.int32)
        b = _batch(data, src_off, src_len, out, dst_off, dst_len, res)
        _lib.check(_lib.lib().lz4hip_decode_batch_host(C.byref(b), 1))
        if not (res == src_len).all():                       # Decode64: consumed != inputLength (Unsafe.cs:373-378)
            raise ArgumentException("LZ4 block is corrupted, or invalid length has been given.")
    return out.tobytes()
