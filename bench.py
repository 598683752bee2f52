#!/usr/bin/env python
"""bench.py -- headline benchmark: batched 64 KiB-block LZ4 decode on MI355X (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--blocks B] [--dist D] [--no-extras]

One "step" = one pass of the batched known-size decoder over the whole resident batch (2^20 blocks of
64 KiB per GPU by default; every buffer is in HBM before the timed region starts).  For N > 1 the driver
launches one process per GPU (torch.distributed.run); blocks are sharded round-robin, there is no
data-path collective, and the batch grows with N (weak scaling).

Prints ONE JSON line (rank 0).  `value` = uncompressed GB/s of the decode step over all ranks;
`roofline` = algorithmic bytes (u_i + c_i + 8 per block) / mean kernel time from HIP events on the
launch stream vs the 8 TB/s HBM3E peak; `cpu_baseline` = the reference's own C (oracle/_ref, built
from original/lz4.c) or, if absent, the CPU restatement, decoding a bounded sample of the same
workload on this box's host cores.  `extras` carries the other distributions and the encoder rates.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6300 GB/s achievable
DIST_NAMES = {0: "D0-zeros", 1: "D1-incompressible", 2: "D2-fuzzer(original/fuzzer.c)", 3: "D3-records"}


def csrc_sha() -> str:
    """Hash of the kernel sources: committed PMC traffic figures are only quoted for the code they were measured on, and the
    library says which sources IT was compiled from (lz4hip_build_id) -- the two must agree before anything is measured."""
    from lz4net_amd import build as hip_build
    return hip_build.csrc_sha()


def committed_traffic(kind: str, dist: int, blocks: int):
    """HBM-side bytes per launch (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes, tools/pmc_traffic.sh)
    from profiles/r05 (or an earlier round)/pmc_traffic.json -- only if that file was produced from EXACTLY these kernel
    sources and this workload; otherwise None (a stale number would be a lie)."""
    sha = csrc_sha()
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        f = os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")
        if not os.path.exists(f):
            continue
        with open(f) as fh:
            d = json.load(fh)
        if d.get("csrc_sha") != sha:
            continue
        e = d.get(kind)
        if not e or e.get("dist") != dist or e.get("blocks") != blocks:
            return None, None
        return int(e["bytes_per_launch"]), (f"profiles/{rnd}/pmc_traffic.json[{kind}] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                            f"separate passes, per launch; csrc {d['csrc_sha']})")
    return None, None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs (default: WORLD_SIZE if a launcher set it, else 1)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=1 << 20, help="64 KiB blocks PER GPU")
    ap.add_argument("--dist", type=int, default=2, help="headline distribution (0..3)")
    ap.add_argument("--seed", type=int, default=20260925)
    ap.add_argument("--no-extras", action="store_true", help="skip the other distributions / encoder timings")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--hc-only", action="store_true", help="of the extras run only the LZ4HC leg (profiling runs)")
    ap.add_argument("--hc-blocks", type=int, default=1 << 18, help="blocks for the LZ4HC extra (SURVEY 8d C4: 2^18; 0 = skip)")
    ap.add_argument("--decoder", choices=["auto", "lane", "wave"], default="auto",
                    help="block->hardware mapping of the decoder (auto = library default)")
    ap.add_argument("--encoder", choices=["auto", "lane", "wave"], default="auto")
    ap.add_argument("--dst-pad", type=int, default=0, help="extra bytes between decoded blocks (stride experiment)")
    ap.add_argument("--early-workspace", type=int, default=0,
                    help="1: make the library build the lane encoder's table slab BEFORE the batch buffers are allocated; 0 (default): when first needed.  "
                         "(A round-4 experiment from when the slab was ONE allocation whose placement decided between 44-46 and 53-54 GB/s; the slab is now "
                         "built from separately allocated chunks whose placement the library measures: DESIGN.md 4.2)")
    ap.add_argument("--verify-budget", type=float, default=45.0,
                    help="seconds of host time for EACH full-corpus encoder check against the CPU reference (0 = skip)")
    return ap.parse_args()


def relaunch_as_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: start N ranks (one per GPU) under torch.distributed.run
    with the same arguments and hand its exit code back.  The driver's own launch line (python -m torch.distributed.run
    --nproc-per-node N ... bench.py --gpus N) sets WORLD_SIZE and never comes here."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


@contextlib.contextmanager
def _stdout_to_stderr():
    sys.stdout.flush()
    fd = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(fd, 1)
        os.close(fd)


def event_ms(fn, torch):
    """Run fn() bracketed by HIP events on the current stream; returns elapsed ms (synchronises)."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b)


class Workload:
    """One distribution's device-resident batch: raw blocks, compressed blocks, lengths, decode target."""

    def __init__(self, torch, batch, dist, seed, first_block, n, block_step=1, dst_pad=0):
        self.torch, self.batch, self.dist, self.n = torch, batch, dist, n
        self.raw = batch.synth(dist, seed, first_block, n, block_step=block_step)
        self.comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
        self.clen = torch.empty(n, dtype=torch.int32, device="cuda")
        self.back = torch.empty((n, batch.BLOCK + dst_pad), dtype=torch.uint8, device="cuda")
        self.used = torch.empty(n, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        # warm-up launches on slivers (module load; the lane encoder's per-device table workspace is allocated by
        # the first batch of >= 49152 blocks), then the timed single-pass encode
        batch.encode(self.raw[:64], batch.BLOCK, self.comp[:64], batch.BOUND, result=self.clen[:64])
        k = min(n, 49152)
        batch.encode(self.raw[:k], batch.BLOCK, self.comp[:k], batch.BOUND, result=self.clen[:k])
        torch.cuda.synchronize()
        self.encode_ms = min(event_ms(lambda: batch.encode(self.raw, batch.BLOCK, self.comp, batch.BOUND, result=self.clen), torch)
                             for _ in range(2))
        self.comp_bytes = int(self.clen.to(torch.int64).sum().item())
        assert bool((self.clen > 0).all()), "encoder reported failure on a block"
        # what the lane encoder's table slab measured when the library built it (the line must say which
        # placement the encoder ran on): G probe steps per second, candidates built, chunks of the one in use
        from lz4net_amd import _lib
        self.slab = {"encoder_slab_rate_Gsteps": _lib.tuning_get("encoder_slab_rate") / 1000.0,
                     "encoder_slab_tried": _lib.tuning_get("encoder_slab_tried"),
                     "encoder_slab_chunks": _lib.tuning_get("encoder_slab_chunks")}

    def decode_step(self):
        self.batch.decode(self.comp, self.clen, self.back, self.batch.BLOCK, known_output_size=True, result=self.used)

    def verify(self):
        ok = bool((self.used == self.clen).all())
        bad = self.batch.count_mismatches(self.raw, self.back, self.batch.BLOCK)
        return ok and bad == 0

    @property
    def raw_bytes(self):
        return self.n * self.batch.BLOCK

    @property
    def algorithmic_bytes(self):           # sum(u_i + c_i + 8)
        return self.raw_bytes + self.comp_bytes + 8 * self.n


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def full_corpus_encoder_check(torch, batch, comp, clen, hc, dist, seed, first, step, budget):
    """EVERY block of the batch against the CPU codec: per-block (compressed length, checksum of the compressed bytes) from
    the GPU rows vs the same two numbers from the CPU reference, which regenerates each block from its seed, compresses it
    and keeps nothing else (oracle/batch.c lz4o_verify_stream; all host cores; stops after `budget` seconds and says how far
    it got).  Outside every timed region.  The reference's bar: src/LZ4.Tests/ConformanceTests.cs:121-133."""
    import numpy as np
    from oracle.oracle import Oracle, Reference
    o = Oracle()
    codec = Reference() if Reference.available() else o
    n = comp.shape[0]
    g_sum = batch.checksum(comp, clen).cpu().numpy().view(np.uint64)
    g_len = clen.cpu().numpy()
    cores = os.cpu_count() or 1
    t0 = time.time()
    done, c_len, c_sum = o.verify_stream(codec, hc, dist, seed, first, step, n, length=batch.BLOCK, threads=cores, budget_seconds=budget)
    dt = time.time() - t0
    bad = np.nonzero((g_len[:done] != c_len[:done]) | (g_sum[:done] != c_sum[:done]))[0]
    return {"blocks_compared": done, "blocks": n, "all_equal": bool(done > 0 and bad.size == 0), "mismatching_blocks": int(bad.size),
            "first_mismatch": int(bad[0]) if bad.size else None, "cpu_seconds": round(dt, 2), "cpu_codec": codec.kind,
            "what": "per block: compressed length and 64-bit checksum of the compressed bytes, GPU rows vs the CPU codec on the regenerated block"}


def cpu_baseline(dist, seed, gpu_comp_sample, sample_blocks, raw=None):
    """The CPU codec on this box's host cores, bounded sample of the same workload (the first `sample_blocks` blocks of the GPU
    batch: copied from it, or regenerated by the CPU twin of the generator).  All three legs the same way -- best of N passes over
    the whole sample, N bounded by time (the method of the reference's own timers: src/LZ4.Tests.Helpers/TimedMethod.cs:66-69,
    original/bench.c:402-443 keep the best of repeated runs).  Also the bench's parity spot check: the GPU's compressed bytes for
    the first blocks must equal the CPU reference's."""
    import numpy as np
    from oracle.oracle import Oracle, Reference
    o = Oracle()
    codec, kind = (Reference(), "reference") if Reference.available() else (o, "port")
    cores = os.cpu_count() or 1
    if raw is None:
        raw = o.gen(dist, seed, 0, sample_blocks)
    else:
        assert raw.shape[0] == sample_blocks and np.array_equal(raw[:8], o.gen(dist, seed, 0, 8)), "sample is not the head of the workload"
    bound = 65536 + 65536 // 255 + 16
    comp = np.zeros((sample_blocks, bound), np.uint8)
    lens = np.full(sample_blocks, 65536, np.int32)
    caps = np.full(sample_blocks, bound, np.int32)

    def best_of(fn, seconds, min_passes=3, max_passes=50):
        best, passes, t0, last = None, 0, time.time(), None
        while passes < min_passes or (time.time() - t0 < seconds and passes < max_passes):
            t, last = fn()
            best = t if best is None else min(best, t)
            passes += 1
        return best, passes, last

    t_enc, enc_passes, clen = best_of(lambda: o.batch(codec, "enc", raw, lens, comp, caps, threads=cores), 6.0)
    parity = None
    if gpu_comp_sample is not None:
        g_comp, g_len = gpu_comp_sample
        k = min(len(g_len), sample_blocks)
        parity = bool((g_len[:k] == clen[:k]).all()) and all(
            np.array_equal(g_comp[i, :clen[i]], comp[i, :clen[i]]) for i in range(k))
    back = np.zeros_like(raw)
    best, passes, res = best_of(lambda: o.batch(codec, "dec", comp, clen, back, lens, threads=cores), 6.0)
    assert (res == clen).all()
    assert np.array_equal(back, raw)
    # LZ4HC on a quarter of the sample (it is ~4x slower per core than the fast encoder), same best-of-N
    hc_blocks = min(max(cores, sample_blocks // 4), sample_blocks)
    hcomp = np.zeros((hc_blocks, bound), np.uint8)
    t_hc, hc_passes, hlen = best_of(lambda: o.batch(codec, "hc", raw[:hc_blocks], lens[:hc_blocks], hcomp, caps[:hc_blocks], threads=cores), 6.0)
    assert (hlen > 0).all()
    return {
        "value": round(sample_blocks * 65536 / best / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": kind,
        "label": ("C oracle stand-in for LZ4pn: the reference's own original/lz4.c + lz4hc.c (the C the LZ4pn C# is generated from), "
                  "built with the flags of its 64-bit native back-end" if kind == "reference" else
                  "C oracle stand-in for LZ4pn: from-scratch C restatement (oracle/lz4_oracle.c)"),
        "cpu_model": cpu_model(),
        "sample": f"{sample_blocks} x 64 KiB {DIST_NAMES[dist]} blocks (first blocks of the GPU batch, {sample_blocks * 65536 >> 20} MiB raw), {cores} threads, "
                  f"every leg best of N whole passes: decode best of {passes}, fast encode best of {enc_passes} "
                  f"({round(sample_blocks * 65536 / t_enc / 1e9, 3)} GB/s), LZ4HC on the first {hc_blocks} blocks best of {hc_passes}",
        "encode_value": round(sample_blocks * 65536 / t_enc / 1e9, 3), "encode_passes": enc_passes,
        "encode_hc_value": round(hc_blocks * 65536 / t_hc / 1e9, 3), "encode_hc_sample_blocks": hc_blocks, "encode_hc_passes": hc_passes,
        "decode_passes": passes,
        "gpu_bytes_equal_cpu_reference": parity,
    }


def main():
    args = parse_args()
    if args.gpus is None:
        # launched as `torchrun --nproc-per-node N bench.py` without --gpus: the launcher's world size is the answer
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
        if args.gpus > 1 and int(os.environ.get("RANK", "0")) == 0:
            print(f"[bench] --gpus not given: adopting WORLD_SIZE={args.gpus} from the launcher", file=sys.stderr)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_as_ranks(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} is running as {world} rank(s): launch it with --nproc-per-node {args.gpus} "
                         f"(or without a launcher, and it starts the ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the codec")
    # (LZ4HIP_BENCH_SHARE_GPU=1: every rank uses cuda:0 -- only for exercising the N>1 code path on a 1-GPU box)
    share_gpu = bool(os.environ.get("LZ4HIP_BENCH_SHARE_GPU"))
    if not share_gpu and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py --gpus {world}: only {torch.cuda.device_count()} device(s) visible "
                         f"(LZ4HIP_BENCH_SHARE_GPU=1 puts every rank on cuda:0 to exercise the code path, not to measure)")
    torch.cuda.set_device(0 if share_gpu else local_rank)
    if world > 1:
        # The data path has no collective (blocks are independent, sharded round-robin); ranks only meet at
        # the timing barriers and to combine three scalars, which gloo does over host memory.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with _stdout_to_stderr():      # gloo announces its connections on stdout; stdout is reserved for the JSON line
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist.barrier()

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if world > 1:
        dist.barrier()
    from lz4net_amd import batch, _lib
    _lib.lib()
    # the binary that is about to be measured must be the one these sources produce (prebuilt .so files travel between boxes)
    lib_id = _lib.lib().lz4hip_build_id().decode()
    if lib_id.split("+")[0] != csrc_sha():
        raise SystemExit(f"bench.py: lz4net_amd/liblz4hip.so was built from csrc {lib_id}, the tree holds {csrc_sha()}: refusing to measure a stale library")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # ---- size the batch to the memory actually free on this GPU ------------------------------------
    n = args.blocks
    free, total = torch.cuda.mem_get_info()
    per_block = 2 * batch.BLOCK + batch.BOUND_STRIDE + 16
    per_block += args.dst_pad
    while n * per_block * 1.03 > free and n > 1024:
        n //= 2
    if world > 1:
        # every rank must run the SAME shard size (weak scaling: blocks_per_gpu is one number): the smallest any rank can hold
        nt = torch.tensor([n], dtype=torch.int64)
        dist.all_reduce(nt, op=dist.ReduceOp.MIN)
        if int(nt.item()) != n:
            print(f"[bench] rank {rank}: {n} blocks would fit here, another rank holds fewer: every rank runs {int(nt.item())}", file=sys.stderr)
        n = int(nt.item())
    # round-robin shard of a global batch of n*world blocks: local block j is global block j*world + rank
    seed = args.seed
    _lib.tuning_set("decoder", args.decoder)
    _lib.tuning_set("encoder", args.encoder)
    if args.early_workspace:
        # (experiment, off by default) The lane encoder keeps one 32 KiB table per resident lane in a slab that the library allocates
        # when a batch first needs it.  A batch of tiny blocks, large enough for the full residency, makes the library allocate
        # it now, before the 192 GB of batch buffers -- which turned out to be the slower placement.
        tiny_n, tiny_len = 1 << 19, 64
        tiny = batch.synth(2, 1, 0, tiny_n, length=tiny_len)
        tiny_out = torch.empty((tiny_n, 96), dtype=torch.uint8, device="cuda")
        batch.encode(tiny, tiny_len, tiny_out, 80)
        torch.cuda.synchronize()
        del tiny, tiny_out
        torch.cuda.empty_cache()
    wl = Workload(torch, batch, args.dist, seed, rank, n, block_step=world, dst_pad=args.dst_pad)
    for _ in range(max(args.warmup, 0)):
        wl.decode_step()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kernel_ms.append(event_ms(wl.decode_step, torch))
    barrier()
    elapsed = time.perf_counter() - t0
    ok = wl.verify()

    tmax = torch.tensor([elapsed], dtype=torch.float64)
    stats = torch.tensor([float(wl.algorithmic_bytes), float(wl.comp_bytes), float(ok)], dtype=torch.float64)
    # which physical device each rank really ran on (PCI bus id): n_gpus counts ranks, distinct_devices says whether they shared
    props = torch.cuda.get_device_properties(torch.cuda.current_device())
    dev_ids = [(os.uname().nodename, str(getattr(props, "uuid", None) or getattr(props, "pci_bus_id", None) or torch.cuda.current_device()))]
    rank_ms = [(rank, round(elapsed / args.steps * 1e3, 3), round(sum(kernel_ms) / len(kernel_ms), 3))]   # (rank, wall ms/step, kernel ms/step)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        gathered = [None] * world
        dist.all_gather_object(gathered, dev_ids[0])
        dev_ids = gathered
        gathered = [None] * world
        dist.all_gather_object(gathered, rank_ms[0])
        rank_ms = gathered
    distinct_devices = len(set(dev_ids))
    if rank == 0 and world > 1:
        # self-diagnosing multi-GPU runs: which device every rank sat on and how long ITS steps took
        for (rk, wall, kern), dev in zip(rank_ms, dev_ids):
            print(f"[bench] rank {rk}: device {dev[1]} on {dev[0]}: {wall} ms/step wall, {kern} ms/step kernel", file=sys.stderr)
        if distinct_devices < world:
            print(f"[bench] WARNING: {world} ranks ran on {distinct_devices} distinct device(s): this is a code-path check, NOT a scaling measurement", file=sys.stderr)
    elapsed = float(tmax.item())
    all_ok = int(stats[2].item()) == world

    # ---- extras: the other distributions (decode) and the encoders, rank 0 at N == 1 only -----------
    extras = {}
    gpu_sample = None
    # cpu_baseline sample: 16 384 blocks (1 GiB raw -- not LLC-resident on a 256 MiB-L3 host) where the host has the cores to get through
    # it in seconds, 2 048 otherwise; the raw blocks are the GPU batch's own first blocks (copied out here, wl is freed below)
    sample_blocks = min(16384 if (os.cpu_count() or 1) >= 32 else 2048, n)
    raw_sample = None
    if rank == 0:
        k = min(64, n)
        gpu_sample = (wl.comp[:k].cpu().numpy(), wl.clen[:k].cpu().numpy())
        if world == 1 and not args.no_cpu:
            raw_sample = wl.raw[:sample_blocks].cpu().numpy()
    head = {
        "decode_GBps": round(wl.raw_bytes / (sum(kernel_ms) / len(kernel_ms) / 1e3) / 1e9, 2),
        "encode_fast_GBps": round(wl.raw_bytes / (wl.encode_ms / 1e3) / 1e9, 2),
        "ratio": round(wl.comp_bytes / wl.raw_bytes, 4), "blocks": n, "roundtrip_ok": ok,
    }
    extras[DIST_NAMES[args.dist]] = head
    alg_bytes_local, mean_kernel_ms = wl.algorithmic_bytes, sum(kernel_ms) / len(kernel_ms)
    enc_roof = {"ms": wl.encode_ms, "alg": wl.algorithmic_bytes, "blocks": n, "slab": wl.slab}      # BASELINE configs[2]
    hc_roof = None                                                               # BASELINE configs[3]
    if world == 1 and not args.no_extras and not args.hc_only and args.decoder == "auto":
        for name in ("lane", "wave"):
            _lib.tuning_set("decoder", name)
            wl.back.zero_()
            wl.decode_step()
            torch.cuda.synchronize()
            t = min(event_ms(wl.decode_step, torch) for _ in range(2))
            head[f"decode_{name}_GBps"] = round(wl.raw_bytes / (t / 1e3) / 1e9, 2)
            head[f"decode_{name}_ok"] = wl.verify()
        _lib.tuning_set("decoder", "auto")
    enc_check = hc_check = None
    if world == 1 and not args.no_cpu and args.verify_budget > 0 and args.encoder == "auto":
        try:
            enc_check = full_corpus_encoder_check(torch, batch, wl.comp, wl.clen, False, args.dist, seed, 0, 1, args.verify_budget)
        except Exception as e:
            enc_check = {"error": repr(e)}
    if world == 1 and not args.no_extras:
        del wl
        torch.cuda.empty_cache()
        for d in ([] if args.hc_only else range(4)):
            if d == args.dist:
                continue
            w = Workload(torch, batch, d, seed, 0, n, dst_pad=args.dst_pad)
            w.decode_step()
            torch.cuda.synchronize()
            ms = [event_ms(w.decode_step, torch) for _ in range(3)]
            ok_default = w.verify()
            alt = {}
            if args.decoder == "auto":          # A/B: the other mapping on the same batch
                for name in ("lane", "wave"):
                    _lib.tuning_set("decoder", name)
                    w.back.zero_()
                    w.decode_step()
                    torch.cuda.synchronize()
                    t = min(event_ms(w.decode_step, torch) for _ in range(2))
                    alt[f"decode_{name}_GBps"] = round(w.raw_bytes / (t / 1e3) / 1e9, 2)
                    alt[f"decode_{name}_ok"] = w.verify()
                _lib.tuning_set("decoder", "auto")
            other_check = None
            if d in (2, 3) and not args.no_cpu and args.verify_budget > 0 and args.encoder == "auto":
                # the OTHER sequence-dense distribution: every block of it against the CPU reference as well
                try:
                    other_check = full_corpus_encoder_check(torch, batch, w.comp, w.clen, False, d, seed, 0, 1, args.verify_budget)
                except Exception as e:
                    other_check = {"error": repr(e)}
            extras[DIST_NAMES[d]] = {
                **alt,
                **({"encode_fast_bit_exact_vs_cpu_reference": other_check} if other_check is not None else {}),
                "decode_GBps": round(w.raw_bytes / (min(ms) / 1e3) / 1e9, 2),
                "decode_frac_of_hbm_peak": round(w.algorithmic_bytes / (min(ms) / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
                "encode_fast_GBps": round(w.raw_bytes / (w.encode_ms / 1e3) / 1e9, 2),
                "ratio": round(w.comp_bytes / w.raw_bytes, 4), "blocks": n, "roundtrip_ok": ok_default,
            }
            del w
            torch.cuda.empty_cache()
        if args.hc_blocks > 0:
            m = min(args.hc_blocks, n)
            free_b, total_b = torch.cuda.mem_get_info()
            print(f"[bench] before LZ4HC: {free_b / 2**30:.1f} GiB free of {total_b / 2**30:.1f}", file=sys.stderr)
            raw = batch.synth(args.dist, seed, 0, m)
            comp = torch.empty((m, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
            batch.encode(raw[:64], batch.BLOCK, comp[:64], batch.BOUND, hc=True)
            torch.cuda.synchronize()
            # two passes: the first one also allocates the per-device LZ4HC workspace (192 KiB per resident lane,
            # a one-time cost per process); the rate quoted is the pass that finds it in place
            ms_all = []
            for _ in range(2):
                clen_holder = {}
                ms_all.append(event_ms(lambda: clen_holder.setdefault("c", batch.encode(raw, batch.BLOCK, comp, batch.BOUND, hc=True)), torch))
            ms = min(ms_all)
            clen = clen_holder["c"]
            back = torch.empty_like(raw)
            used = batch.decode(comp, clen, back, batch.BLOCK)
            hc_roof = {"ms": ms, "alg": m * batch.BLOCK + int(clen.to(torch.int64).sum().item()) + 8 * m, "blocks": m}
            if not args.no_cpu and args.verify_budget > 0:
                try:
                    hc_check = full_corpus_encoder_check(torch, batch, comp, clen, True, args.dist, seed, 0, 1, args.verify_budget)
                except Exception as e:
                    hc_check = {"error": repr(e)}
            extras["LZ4HC " + DIST_NAMES[args.dist]] = {
                "encode_hc_GBps": round(m * batch.BLOCK / (ms / 1e3) / 1e9, 3),
                "ratio": round(float(clen.double().sum().item()) / (m * batch.BLOCK), 4), "blocks": m,
                "first_pass_with_workspace_allocation_GBps": round(m * batch.BLOCK / (ms_all[0] / 1e3) / 1e9, 3),
                "roundtrip_ok": bool((used == clen).all()) and batch.count_mismatches(raw, back, batch.BLOCK) == 0,
            }
            del raw, comp, back
            if not args.hc_only:
                # the other sequence-dense distribution: round trip AND every block against the CPU reference
                other = 3 if args.dist == 2 else 2
                torch.cuda.empty_cache()
                raw = batch.synth(other, seed, 0, m)
                comp = torch.empty((m, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
                holder = {}
                ms_o = min(event_ms(lambda: holder.__setitem__("c", batch.encode(raw, batch.BLOCK, comp, batch.BOUND, hc=True)), torch) for _ in range(2))
                back = torch.empty_like(raw)
                used = batch.decode(comp, holder["c"], back, batch.BLOCK)
                hc_other_check = None
                if not args.no_cpu and args.verify_budget > 0:
                    try:
                        hc_other_check = full_corpus_encoder_check(torch, batch, comp, holder["c"], True, other, seed, 0, 1, args.verify_budget)
                    except Exception as e:
                        hc_other_check = {"error": repr(e)}
                extras["LZ4HC " + DIST_NAMES[other]] = {
                    "bit_exact_vs_cpu_reference": hc_other_check,
                    "encode_hc_GBps": round(m * batch.BLOCK / (ms_o / 1e3) / 1e9, 3),
                    "ratio": round(float(holder["c"].double().sum().item()) / (m * batch.BLOCK), 4), "blocks": m,
                    "roundtrip_ok": bool((used == holder["c"]).all()) and batch.count_mismatches(raw, back, batch.BLOCK) == 0,
                }
                del raw, comp, back
        if not args.hc_only:
            # ---- what smaller batches get (device-resident, default dispatch): the lane mappings need the chip full, below
            #      16 384 (decode) / 49 152 (fast encode) blocks the wavefront mappings run alone (their time is one wavefront's instruction count) ----
            torch.cuda.empty_cache()
            sweep = {}
            m_max = min(1 << 18, n)
            raw_s = batch.synth(args.dist, seed, 0, m_max)
            comp_s = torch.empty((m_max, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
            clen_s = batch.encode(raw_s, batch.BLOCK, comp_s, batch.BOUND)
            back_s = torch.empty_like(raw_s)
            last_m = 0
            comp_h = None
            for m in (1024, 4096, 16384, 65536, 262144):
                if m > m_max:
                    continue
                last_m = m
                batch.decode(comp_s[:m], clen_s[:m], back_s[:m], batch.BLOCK)
                torch.cuda.synchronize()
                t_dec = min(event_ms(lambda: batch.decode(comp_s[:m], clen_s[:m], back_s[:m], batch.BLOCK), torch) for _ in range(3))
                entry = {"decode_GBps": round(m * batch.BLOCK / (t_dec / 1e3) / 1e9, 2)}
                if m <= 65536:
                    t_enc = min(event_ms(lambda: batch.encode(raw_s[:m], batch.BLOCK, comp_s[:m], batch.BOUND), torch) for _ in range(2))
                    entry["encode_fast_GBps"] = round(m * batch.BLOCK / (t_enc / 1e3) / 1e9, 2)
                    if comp_h is None:
                        comp_h = torch.empty((min(65536, m_max), batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
                    batch.encode(raw_s[:m], batch.BLOCK, comp_h[:m], batch.BOUND, hc=True)      # (untimed: workspace, first launch)
                    torch.cuda.synchronize()
                    t_hc = event_ms(lambda: batch.encode(raw_s[:m], batch.BLOCK, comp_h[:m], batch.BOUND, hc=True), torch)
                    entry["encode_hc_GBps"] = round(m * batch.BLOCK / (t_hc / 1e3) / 1e9, 2)
                sweep[str(m)] = entry
            sweep["ok"] = last_m > 0 and batch.count_mismatches(raw_s[:last_m], back_s[:last_m], batch.BLOCK) == 0
            extras["batch_size_sweep_" + DIST_NAMES[args.dist].split("(")[0]] = sweep
            del raw_s, comp_s, back_s, comp_h
            # ---- the wavefront-mapped fast encoder on its own (what every batch below 49 152 blocks and every host-pointer slice runs; second
            #      version since round 6): 65 536 blocks of both sequence-dense distributions, forced mapping, EVERY block against the CPU reference ----
            wave_enc = {}
            for d in (2, 3):
                torch.cuda.empty_cache()
                m = min(65536, n)
                raw_w = batch.synth(d, seed, 0, m)
                comp_w = torch.empty((m, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
                with _lib.tuning(encoder="wave"):
                    clen_w = batch.encode(raw_w, batch.BLOCK, comp_w, batch.BOUND)
                    torch.cuda.synchronize()
                    hold = {}
                    t_w = min(event_ms(lambda: hold.__setitem__("c", batch.encode(raw_w, batch.BLOCK, comp_w, batch.BOUND)), torch) for _ in range(2))
                    clen_w = hold["c"]
                chk = None
                if not args.no_cpu and args.verify_budget > 0:
                    try:
                        chk = full_corpus_encoder_check(torch, batch, comp_w, clen_w, False, d, seed, 0, 1, args.verify_budget)
                    except Exception as e:
                        chk = {"error": repr(e)}
                wave_enc[DIST_NAMES[d]] = {"encode_fast_GBps": round(m * batch.BLOCK / (t_w / 1e3) / 1e9, 2), "blocks": m, "bit_exact_vs_cpu_reference": chk}
                del raw_w, comp_w
            extras["wavefront_encoder_forced"] = wave_enc
            torch.cuda.empty_cache()
            # ---- the HBM roof measured on this box (quote the 8 TB/s spec AND what a copy reaches) ----
            torch.cuda.empty_cache()
            words = (4 << 30) // 8
            xa = torch.empty(words, dtype=torch.int64, device="cuda")
            xb = torch.empty(words, dtype=torch.int64, device="cuda")
            xa.fill_(1); xb.copy_(xa); torch.cuda.synchronize()
            t_fill = min(event_ms(lambda: xa.fill_(2), torch) for _ in range(3))
            t_copy = min(event_ms(lambda: xb.copy_(xa), torch) for _ in range(3))
            extras["hbm_measured"] = {
                "fill_GBps_write_only": round(words * 8 / (t_fill / 1e3) / 1e9, 1),
                "copy_GBps_read_plus_write": round(2 * words * 8 / (t_copy / 1e3) / 1e9, 1),
                "bytes": words * 8, "note": "torch fill_/copy_ of a 4 GiB buffer, best of 3 (the spec-sheet 8000 GB/s is what roofline.peak quotes)",
            }
            del xa, xb
            torch.cuda.empty_cache()
            # ---- PCIe-inclusive rate of the host-pointer batch entry point (never `value`) ----
            import ctypes as C
            import numpy as np

            def host_decode_rate(m, multi=False):
                L = _lib.lib()
                dec_call = (lambda b: L.lz4hip_decode_batch_host_multi(C.byref(b), 1, 0)) if multi else (lambda b: L.lz4hip_decode_batch_host(C.byref(b), 1))
                enc_call = (lambda b: L.lz4hip_encode_batch_host_multi(C.byref(b), 0, 0)) if multi else (lambda b: L.lz4hip_encode_batch_host(C.byref(b), 0))
                raw_d = batch.synth(args.dist, seed, 0, m)
                comp_d = torch.empty((m, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
                clen_h = batch.encode(raw_d, batch.BLOCK, comp_d, batch.BOUND).cpu().numpy().astype(np.int32)
                comp_h, raw_h = comp_d.cpu().numpy(), raw_d.cpu().numpy()
                del raw_d, comp_d
                back_h = np.zeros_like(raw_h)
                caps_h = np.full(m, batch.BLOCK, np.int32)
                res_h = np.zeros(m, np.int32)
                hb = _lib.Batch(src=comp_h.ctypes.data, src_off=None, src_stride=comp_h.strides[0], src_len=clen_h.ctypes.data,
                                dst=back_h.ctypes.data, dst_off=None, dst_stride=back_h.strides[0], dst_cap=caps_h.ctypes.data,
                                dst_cap_all=0, src_len_all=0, result=res_h.ctypes.data, n_blocks=m)
                _lib.check(dec_call(hb))
                t_host = None
                for _ in range(3):
                    t1 = time.perf_counter()
                    _lib.check(dec_call(hb))
                    dt = time.perf_counter() - t1
                    t_host = dt if t_host is None else min(t_host, dt)
                ok_h = bool((res_h == clen_h).all()) and bool(np.array_equal(back_h, raw_h))
                # ... and the encode side of the same entry points: what a caller shaped like LZ4Stream.FlushCurrentChunk
                # (reference src/LZ4/LZ4Stream.cs:239-269) gets -- pageable rows in, compressed rows out
                enc_h = np.zeros_like(comp_h)
                ecap_h = np.full(m, batch.BOUND, np.int32)
                elen_h = np.full(m, batch.BLOCK, np.int32)
                eres_h = np.zeros(m, np.int32)
                eb = _lib.Batch(src=raw_h.ctypes.data, src_off=None, src_stride=raw_h.strides[0], src_len=elen_h.ctypes.data,
                                dst=enc_h.ctypes.data, dst_off=None, dst_stride=enc_h.strides[0], dst_cap=ecap_h.ctypes.data,
                                dst_cap_all=0, src_len_all=batch.BLOCK, result=eres_h.ctypes.data, n_blocks=m)
                _lib.check(enc_call(eb))
                t_enc = None
                for _ in range(3):
                    t1 = time.perf_counter()
                    _lib.check(enc_call(eb))
                    dt = time.perf_counter() - t1
                    t_enc = dt if t_enc is None else min(t_enc, dt)
                ok_e = bool((eres_h == clen_h).all()) and all(np.array_equal(enc_h[i, :clen_h[i]], comp_h[i, :clen_h[i]]) for i in range(0, m, max(m // 256, 1)))
                return round(m * batch.BLOCK / t_host / 1e9, 2), ok_h, round(m * batch.BLOCK / t_enc / 1e9, 2), ok_e

            m_big, m_small = min(16384, n), min(4096, n)
            rate_big, ok_big, erate_big, eok_big = host_decode_rate(m_big)
            rate_small, ok_small, erate_small, eok_small = host_decode_rate(m_small)
            extras["host_pointer_batch_pcie_inclusive"] = {
                "decode_GBps": rate_big, "blocks": m_big, "decode_GBps_small_batch": rate_small, "blocks_small_batch": m_small,
                "encode_fast_GBps": erate_big, "encode_fast_GBps_small_batch": erate_small,
                "ok": ok_big and ok_small and eok_big and eok_small,
                "note": "lz4hip_decode_batch_host on pageable host arrays: gather + H2D + kernels + D2H + scatter, best of 3 "
                        "(reported beside, never as, `value`); a batch is cut into 1-6 slices (one per ~2048 blocks) whose copies and kernels overlap",
            }
            # ---- the same call through the C ABI's OWN sharding (lz4hip_*_batch_host_multi, device_mask 0 = every visible device: block i -> device
            #      i mod N, one persistent worker per device): the single-process form of BASELINE configs[4].  On a one-GPU box the knob
            #      logical_devices makes two workers share the device -- a recorded rate of the code path, not a scaling point ----
            ndev = _lib.lib().lz4hip_device_count()
            logical = 0 if ndev > 1 else 2
            with _lib.tuning(logical_devices=logical):
                mrate, mok, merate, meok = host_decode_rate(m_big, multi=True)
            extras["host_pointer_batch_multi_device"] = {
                "decode_GBps": mrate, "encode_fast_GBps": merate, "blocks": m_big, "visible_devices": ndev,
                "device_workers": ndev if ndev > 1 else logical, "ok": mok and meok,
                "note": "lz4hip_decode/encode_batch_host_multi, device_mask 0, round-robin shards, results in global order; with one visible device two "
                        "workers share it (knob logical_devices): the threaded path's rate, not a multi-GPU measurement",
            }

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    cpu = None
    if world == 1 and not args.no_cpu:
        try:
            cpu = cpu_baseline(args.dist, seed, gpu_sample, sample_blocks, raw_sample)
        except Exception as e:   # the bench line must still be printed
            cpu = {"error": repr(e)}

    # HBM-side traffic of the kernels: PMC counters cannot be collected inside this process, so they come from the
    # committed rocprofv3 --pmc summary (profiles/r02/pmc_traffic.json, tools/pmc_traffic.sh) -- keyed on a hash of
    # lz4net_amd/csrc/ and on the workload; null when either differs.
    traffic, traffic_src = (None, None)
    if world == 1 and args.decoder == "auto":
        traffic, traffic_src = committed_traffic("decode", args.dist, n)
    total_raw = n * batch.BLOCK * world
    ms_per_step = elapsed / args.steps * 1e3
    achieved = alg_bytes_local / (mean_kernel_ms / 1e3) / 1e9
    wave_mapped = args.decoder == "wave" or (args.decoder == "auto" and (head["ratio"] < 0.125 or head["ratio"] > 0.9))

    def side_roofline(r, kind, kernel, check=None):
        if r is None:
            return None
        a = r["alg"] / (r["ms"] / 1e3) / 1e9
        t, tsrc = committed_traffic(kind, args.dist, r["blocks"]) if world == 1 else (None, None)
        return {"bound": "hbm", "kernel": kernel, "achieved": round(a, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(a / HBM_PEAK_GBS, 5), "traffic": t, "traffic_source": tsrc,
                "algorithmic_bytes_per_launch": r["alg"], "kernel_ms": round(r["ms"], 3), "blocks": r["blocks"],
                "uncompressed_GBps": round(r["blocks"] * batch.BLOCK / (r["ms"] / 1e3) / 1e9, 2),
                **(r.get("slab") or {}),
                "bit_exact_vs_cpu_reference": check}

    line = {
        "metric": "uncompressed GB/s, batched 64KiB-block encode+decode: value = decode (known output size) per step over the "
                  "whole resident batch; fast encode and LZ4HC encode of the same data in roofline_encode / roofline_hc",
        "value": round(total_raw / (elapsed / args.steps) / 1e9, 2),
        "unit": "GB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[1]: batched decode of {n} x 64 KiB pre-compressed blocks per GPU "
                        f"({DIST_NAMES[args.dist]}, compressed on the GPU by the bit-exact fast encoder), known output size",
            "blocks_per_gpu": n, "block_bytes": batch.BLOCK, "distribution": DIST_NAMES[args.dist],
            "sharding": f"round-robin by rank, {world} rank(s), no data-path collective",
            "ranks": world, "distinct_devices": distinct_devices,
            "per_rank_ms_per_step": {"min": min(r[1] for r in rank_ms), "max": max(r[1] for r in rank_ms),
                                     "kernel_min": min(r[2] for r in rank_ms), "kernel_max": max(r[2] for r in rank_ms)},
            "valid_scaling_point": distinct_devices == world,
            "frac_of_aggregate_hbm_peak": round(float(stats[0].item()) / (elapsed / args.steps) / 1e9 / (HBM_PEAK_GBS * world), 4),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": ("lz4hip::decode_kernel<true> (one wavefront per block)" if wave_mapped
                       else "lz4hip::decode_lane4_kernel<true,192,32,128,2,2,2,16> (one lane per block: input window in registers fed from whole 64-byte sectors, 192-byte LDS output ring with rows stored twice instead of wrapped, 128-byte flush units, hand-counted vmcnt)"),
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": alg_bytes_local, "mean_kernel_ms": round(mean_kernel_ms, 4),
        },
        # BASELINE configs[2] / [3]: one launch over the batch, HIP events on the launch stream, same algorithmic bytes
        "roofline_encode": side_roofline(enc_roof, "encode_fast", "lz4hip::encode_fast_kernel (one wavefront per block, 64-probe search, hands dense blocks over) + "
                                         "lz4hip::encode_fast_lane_kernel (LZ4_compress64kCtx, one lane per block, the blocks handed over)", enc_check),
        "roofline_hc": side_roofline(hc_roof, "encode_hc", "lz4hip::hc_nat_chain_kernel + hc_lcp_fill_kernel + encode_hc_lcp_kernel (LZ4_compressHCCtx: chains and shared lengths of every position built up front, then one lane per block, convergent state machine without the insert loop)", hc_check),
        "cpu_baseline": cpu,
        "verified": all_ok,
        "csrc_sha": csrc_sha(), "library_build_id": lib_id,
        "extras": extras,
    }
    print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
