"""Legacy LZ4 command-line frame of the reference era, batched.

Wire format of the upstream demo CLI that ships with the reference (original/lz4demo.c):

    LE32 magic 0x184C2102 (:86)  { LE32 compressedSize  payload }*

The writer (compress_file, :166-249) cuts the input into 8 MiB chunks (CHUNKSIZE, :84), compresses each with
``LZ4_compress`` (or ``LZ4_compressHC``) into an ``LZ4_compressBound`` buffer and prefixes it with its size.
The reader (decode_file, :252-317) checks the magic, then per chunk reads the size -- a size equal to the magic
means "another frame was appended, keep going" (:289-290) -- and decodes with
``LZ4_uncompress_unknownOutputSize(in, out, size, CHUNKSIZE)``; a negative result is a corrupted file.

The reference handles one chunk per call; here all chunks of a buffer go through ONE
lz4hip_encode_batch_host / lz4hip_decode_batch_host call (include/lz4hip.h).  8 MiB chunks are above 64 KiB, so
the encoder takes the generic variant (``LZ4_compressCtx``, original/lz4.c:345-562) exactly like the reference.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .codec import ArgumentException
from .stream import _batch

MAGIC = 0x184C2102
CHUNK_SIZE = 8 << 20


def _bound(n: int) -> int:
    return n + n // 255 + 16                                       # LZ4_compressBound, original/lz4.h:85-86


def compress_frame(data, high_compression: bool = False, chunk_size: int = CHUNK_SIZE) -> bytes:
    """compress_file(): magic + size-prefixed chunks, every chunk compressed in one GPU batch."""
    raw = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    out = bytearray(MAGIC.to_bytes(4, "little"))
    n = (raw.size + chunk_size - 1) // chunk_size
    if n == 0:
        return bytes(out)
    offs = np.arange(n, dtype=np.int64) * chunk_size
    lens = np.minimum(chunk_size, raw.size - offs).astype(np.int32)
    caps = np.array([_bound(int(l)) for l in lens], dtype=np.int32)
    dst_off = np.concatenate(([0], np.cumsum(caps[:-1], dtype=np.int64))).astype(np.int64)
    comp = np.zeros(int(caps.astype(np.int64).sum()), dtype=np.uint8)
    res = np.zeros(n, dtype=np.int32)
    b = _batch(raw, offs, lens, comp, dst_off, caps, res)
    _lib.check(_lib.lib().lz4hip_encode_batch_host(C.byref(b), _lib.MODE_HC if high_compression else _lib.MODE_FAST))
    for i in range(n):
        size = int(res[i])
        if size <= 0:                                              # cannot happen with a compressBound-sized buffer
            raise ArgumentException("LZ4 compression failed")
        out += size.to_bytes(4, "little")
        out += comp[int(dst_off[i]):int(dst_off[i]) + size].tobytes()
    return bytes(out)


def parse_frame(frame):
    """Header walk -> list of (payload_offset, payload_length); raises on a bad magic or a truncated chunk."""
    buf = memoryview(frame)
    if len(buf) < 4 or int.from_bytes(buf[:4], "little") != MAGIC:
        raise ArgumentException("Unrecognized header : file cannot be decoded")
    pos, chunks = 4, []
    while pos < len(buf):
        if pos + 4 > len(buf):
            raise ArgumentException("truncated chunk header")
        size = int.from_bytes(buf[pos:pos + 4], "little")
        pos += 4
        if size == MAGIC:                                          # appended compressed stream (lz4demo.c:289-290)
            continue
        if pos + size > len(buf):
            raise ArgumentException("truncated chunk payload")
        chunks.append((pos, size))
        pos += size
    return chunks


def decompress_frame(frame, chunk_size: int = CHUNK_SIZE) -> bytes:
    """decode_file(): every chunk decoded (output size unknown, at most `chunk_size`) in one GPU batch."""
    data = np.frombuffer(bytes(frame), dtype=np.uint8)
    chunks = parse_frame(memoryview(bytes(frame)))
    n = len(chunks)
    if n == 0:
        return b""
    src_off = np.array([c[0] for c in chunks], dtype=np.int64)
    src_len = np.array([c[1] for c in chunks], dtype=np.int32)
    dst_off = np.arange(n, dtype=np.int64) * chunk_size
    caps = np.full(n, chunk_size, dtype=np.int32)
    out = np.zeros(n * chunk_size, dtype=np.uint8)
    res = np.zeros(n, dtype=np.int32)
    b = _batch(data, src_off, src_len, out, dst_off, caps, res)
    _lib.check(_lib.lib().lz4hip_decode_batch_host(C.byref(b), 0))
    if (res < 0).any():
        raise ArgumentException("Decoding Failed ! Corrupted input !")
    return b"".join(out[int(dst_off[i]):int(dst_off[i]) + int(res[i])].tobytes() for i in range(n))
