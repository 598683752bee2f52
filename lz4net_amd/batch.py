"""Device-resident batches of independent LZ4 blocks on torch tensors, and round-robin sharding of a
batch across the GPUs of a node.

torch is plumbing here (device memory, streams, torch.distributed); every byte of codec work happens
in the HIP kernels behind lz4hip_encode_batch_device / lz4hip_decode_batch_device (include/lz4hip.h),
launched on torch's current stream.

Multi-GPU: blocks are independent (the reference allocates a fresh table per call,
original/lz4.c:583,780), so block i belongs to rank i % world_size, each rank keeps its own
src/dst/length arrays, and NO collective touches payload bytes.  Only the 4-byte per-block results
are gathered, on the host side.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

BLOCK = 65536
BOUND = BLOCK + BLOCK // 255 + 16          # MaximumOutputLength(65536) = 65809
BOUND_STRIDE = (BOUND + 15) // 16 * 16     # 65824: 16-byte aligned slot for one compressed block


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def _lens(x, n, device):
    """int -> (None, value); tensor -> (int32 device tensor, upper-bound hint 0)."""
    if isinstance(x, int):
        return None, x
    assert x.dtype == torch.int32 and x.numel() == n and x.device.type == "cuda"
    return x, 0


def _make_batch(src, src_len, dst, dst_cap, result, src_len_hint=0):
    assert src.dtype == torch.uint8 and dst.dtype == torch.uint8 and src.dim() == 2 and dst.dim() == 2
    assert src.is_cuda and dst.is_cuda and src.stride(1) == 1 and dst.stride(1) == 1
    n = src.shape[0]
    assert dst.shape[0] == n and result.numel() == n and result.dtype == torch.int32
    sl, sl_all = _lens(src_len, n, src.device)
    dc, dc_all = _lens(dst_cap, n, src.device)
    if sl is not None:
        sl_all = src_len_hint
    b = _lib.Batch(src=src.data_ptr(), src_off=None, src_stride=src.stride(0), src_len=_ptr(sl),
                   dst=dst.data_ptr(), dst_off=None, dst_stride=dst.stride(0), dst_cap=_ptr(dc),
                   dst_cap_all=dc_all, src_len_all=sl_all, result=result.data_ptr(), n_blocks=n)
    return b, (sl, dc)      # keep the tensors alive until the launch has been enqueued


def encode(src: torch.Tensor, src_len, dst: torch.Tensor, dst_cap, hc: bool = False,
           result: torch.Tensor | None = None, src_len_hint: int = 0) -> torch.Tensor:
    """Compress row i of `src` (src_len bytes) into row i of `dst` (capacity dst_cap).  Returns the
    int32 per-block results (bytes written, 0 = did not fit) -- LZ4_compress[HC]_limitedOutput semantics."""
    if result is None:
        result = torch.empty(src.shape[0], dtype=torch.int32, device=src.device)
    b, keep = _make_batch(src, src_len, dst, dst_cap, result, src_len_hint)
    _lib.check(_lib.lib().lz4hip_encode_batch_device(C.byref(b), _lib.MODE_HC if hc else _lib.MODE_FAST, _stream()))
    return result


def decode(src: torch.Tensor, src_len, dst: torch.Tensor, out_size, known_output_size: bool = True,
           result: torch.Tensor | None = None) -> torch.Tensor:
    """Decompress row i of `src`.  known_output_size=True: out_size is the exact decoded size and the
    result is the number of source bytes consumed (LZ4_uncompress); False: out_size is a capacity and
    the result is the number of bytes produced (LZ4_uncompress_unknownOutputSize).  Negative = error."""
    if result is None:
        result = torch.empty(src.shape[0], dtype=torch.int32, device=src.device)
    b, keep = _make_batch(src, src_len, dst, out_size, result)
    _lib.check(_lib.lib().lz4hip_decode_batch_device(C.byref(b), 1 if known_output_size else 0, _stream()))
    return result


def synth(dist: int, seed: int, first_block: int, n_blocks: int, length: int = BLOCK, stride: int | None = None,
          out: torch.Tensor | None = None, device=None, block_step: int = 1) -> torch.Tensor:
    """Synthetic blocks generated on the device (bit-identical to oracle/synth.c); row i is synthetic block
    first_block + i * block_step, so (first_block=rank, block_step=world) is a rank's round-robin share."""
    stride = length if stride is None else stride
    if out is None:
        out = torch.empty((n_blocks, stride), dtype=torch.uint8, device=device or torch.device("cuda"))
    _lib.check(_lib.lib().lz4hip_synth_device(dist, seed, first_block, block_step, n_blocks, out.data_ptr(), out.stride(0),
                                              length, _stream()))
    return out


def checksum(data: torch.Tensor, lens) -> torch.Tensor:
    """Per-block 64-bit checksums (returned as int64 bit patterns)."""
    n = data.shape[0]
    sums = torch.empty(n, dtype=torch.int64, device=data.device)
    ln, ln_all = _lens(lens, n, data.device)
    _lib.check(_lib.lib().lz4hip_checksum_device(data.data_ptr(), None, data.stride(0), _ptr(ln), ln_all,
                                                 sums.data_ptr(), n, _stream()))
    return sums


def count_mismatches(a: torch.Tensor, b: torch.Tensor, lens) -> int:
    """Number of differing bytes between rows of a and b (synchronises)."""
    n = a.shape[0]
    bad = torch.zeros(1, dtype=torch.int64, device=a.device)
    ln, ln_all = _lens(lens, n, a.device)
    _lib.check(_lib.lib().lz4hip_compare_device(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), _ptr(ln), ln_all,
                                                n, bad.data_ptr(), _stream()))
    return int(bad.item())


# ---- round-robin sharding -------------------------------------------------------------------------
def local_block_count(n_blocks: int, rank: int, world: int) -> int:
    """Blocks owned by `rank` when block i lives on rank i % world."""
    return (n_blocks - rank + world - 1) // world if n_blocks > rank else 0


def local_to_global(j, rank: int, world: int):
    return j * world + rank


def gather_results(local: torch.Tensor, n_blocks: int, group=None) -> torch.Tensor | None:
    """Host-side gather of the per-block int32 results of a round-robin sharded batch, in global block
    order, on rank 0 (None elsewhere).  This is metadata (4 B per block); payloads stay where they are."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local.cpu()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (n_blocks + world - 1) // world
    pad = torch.full((per,), -(2 ** 31), dtype=torch.int32)
    pad[:local.numel()] = local.cpu()
    parts = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, parts, dst=0, group=group)
    if rank != 0:
        return None
    out = torch.stack(parts, dim=1).reshape(-1)[:n_blocks]     # [j, r] -> global index j*world + r
    return out.contiguous()
