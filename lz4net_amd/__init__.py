"""lz4net_amd -- MI355X (gfx950) batched LZ4 block codec behind lz4net's LZ4Codec API.

  lz4net_amd.LZ4Codec      host-side mirror of LZ4.LZ4Codec (src/LZ4/LZ4Codec.cs) over the C ABI
  lz4net_amd.batch         device-resident batches on torch tensors (+ round-robin multi-GPU sharding)
  lz4net_amd.stream        LZ4Stream chunk framing with all chunks of a buffer in one GPU batch
  lz4net_amd._lib          ctypes binding of liblz4hip.so (include/lz4hip.h)

The codec itself is hand-written HIP (lz4net_amd/csrc); Python only moves pointers.
"""
from ._lib import Lz4HipError, lib  # noqa: F401
from .codec import LZ4Codec  # noqa: F401

__all__ = ["LZ4Codec", "Lz4HipError", "lib"]
