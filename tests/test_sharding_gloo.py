"""N>1 path on CPU: two gloo processes shard a batch round-robin (block i -> rank i % world), and the
host-side gather of per-block results restores global order.  No GPU, no payload collective."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_blocks, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lz4net_amd import batch
    cnt = batch.local_block_count(n_blocks, rank, world)
    # stand-in for the per-block results a rank's kernels would produce: f(global index)
    local = torch.tensor([1000 + 3 * batch.local_to_global(j, rank, world) for j in range(cnt)], dtype=torch.int32)
    got = batch.gather_results(local, n_blocks)
    if rank == 0:
        out.put(got.tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_round_robin_gather():
    ctx = mp.get_context("spawn")
    for n_blocks in (1, 2, 7, 64, 65):
        q = ctx.Queue()
        port = 29500 + (os.getpid() + n_blocks) % 2000
        procs = [ctx.Process(target=_worker, args=(r, 2, port, n_blocks, q)) for r in range(2)]
        for p in procs:
            p.start()
        got = q.get(timeout=120)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        assert got == [1000 + 3 * i for i in range(n_blocks)], n_blocks
