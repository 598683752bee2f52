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


def _gpu_worker(rank, world, port, n_blocks, out):
    """One rank of the real N>1 path: its round-robin share is generated, compressed and decompressed on ITS GPU
    (rank % device_count: two ranks share the device on a 1-GPU box), only the 4-byte results are gathered."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(rank % torch.cuda.device_count())
    from lz4net_amd import batch
    cnt = batch.local_block_count(n_blocks, rank, world)
    raw = batch.synth(2, 4242, rank, cnt, block_step=world)          # local block j == global block j*world + rank
    comp = torch.empty((cnt, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
    back = torch.empty_like(raw)
    used = batch.decode(comp, clen, back, batch.BLOCK)
    torch.cuda.synchronize()
    ok = bool((used == clen).all()) and batch.count_mismatches(raw, back, batch.BLOCK) == 0
    sums = batch.checksum(back, batch.BLOCK)
    got = batch.gather_results(clen, n_blocks)
    got_sums = batch.gather_results((sums & 0x7FFFFFFF).to(torch.int32), n_blocks)
    flags = [None] * world
    dist.all_gather_object(flags, ok)
    if rank == 0:
        out.put((got.tolist(), got_sums.tolist(), flags))
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_ranks_run_the_codec_and_gather(oracle):
    n_blocks = 131
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 977) % 2000
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, n_blocks, q)) for r in range(2)]
    for p in procs:
        p.start()
    lens, sums, flags = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert flags == [True, True]
    for i in range(n_blocks):                                        # global order, whichever rank did the work
        block = oracle.gen(2, 4242, i, 1)[0]
        assert lens[i] == len(oracle.compress(block)), i
        assert sums[i] == (oracle.checksum(block) & 0x7FFFFFFF), i


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` with N > 1 and no launcher re-executes itself under torch.distributed.run with
    --nproc-per-node N on 127.0.0.1 and the same arguments; under a launcher (WORLD_SIZE set) it does not, and a world
    size that differs from --gpus is refused instead of silently benchmarking one GPU."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert env.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    # under a launcher with the wrong world size: refused
    calls.clear()
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert not calls and "--gpus 4" in str(e.value.code) and "1 rank" in str(e.value.code)


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher():
    """On the GPU box: `LZ4HIP_BENCH_SHARE_GPU=1 python bench.py --gpus 2` (no launcher) runs two ranks -- both on cuda:0 when
    the box has one GPU -- and prints ONE JSON line with n_gpus 2, the round-robin shards verified on both ranks."""
    import json
    import subprocess
    env = dict(os.environ, LZ4HIP_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--blocks", "32768", "--steps", "2",
                        "--warmup", "1", "--no-extras", "--no-cpu"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["verified"] is True and d["config"]["ranks"] == 2
    assert d["config"]["distinct_devices"] in (1, 2)
    assert d["config"]["blocks_per_gpu"] == 32768 and d["value"] > 0
