"""-m gpu: device-resident batches (torch tensors -> lz4hip_*_batch_device), device-side synthetic
generators, and the size-independent properties used at BASELINE.json's full batch sizes:
encode -> decode round trip is the identity, results == lengths, checksum of checksums."""
import numpy as np
import pytest

from conftest import ForcedMapping

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "no GPU visible to torch"
    torch.cuda.set_device(0)
    return torch


def test_device_generators_match_cpu_twins(torch_cuda, oracle):
    from lz4net_amd import batch
    for dist in range(4):
        for length in (1, 100, 4096, 65536):
            got = batch.synth(dist, 77, 1000, 5, length).cpu().numpy()
            want = oracle.gen(dist, 77, 1000, 5, length)
            assert np.array_equal(got, want), (dist, length)
    t = batch.synth(2, 3, 10, 7, 65536)
    sums = batch.checksum(t, 65536).cpu().numpy().view(np.uint64)
    host = t.cpu().numpy()
    assert [int(s) for s in sums] == [oracle.checksum(host[i]) for i in range(7)]


@pytest.mark.parametrize("dist", [0, 1, 2, 3])
@pytest.mark.parametrize("hc", [False, True])
@pytest.mark.parametrize("decoder", ["wave", "lane"])
def test_device_roundtrip_sampled_against_oracle(torch_cuda, oracle, dist, hc, decoder):
    torch = torch_cuda
    from lz4net_amd import batch
    n = 4096 if not hc else 512
    raw = batch.synth(dist, 2024, 0, n)
    comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND, hc=hc)
    back = torch.empty_like(raw)
    with ForcedMapping("LZ4HIP_DECODER", decoder):
        used = batch.decode(comp, clen, back, batch.BLOCK, known_output_size=True)
        assert bool((clen > 0).all()) and bool((used == clen).all())
        assert batch.count_mismatches(raw, back, batch.BLOCK) == 0
        produced = batch.decode(comp, clen, back.zero_(), batch.BLOCK, known_output_size=False)
        assert bool((produced == batch.BLOCK).all()) and batch.count_mismatches(raw, back, batch.BLOCK) == 0
    # spot-check compressed bytes against the oracle on a deterministic sample
    lens = clen.cpu().numpy()
    for i in list(range(0, n, max(n // 16, 1))) + [n - 1]:
        want = oracle.compress(oracle.gen(dist, 2024, i, 1)[0], hc=hc)
        got = comp[i, :int(lens[i])].cpu().numpy()
        assert lens[i] == len(want) and np.array_equal(got, want), (dist, hc, i)


def _full_size(torch, want, bytes_per_block):
    """Blocks for a test that claims BASELINE.json's full batch size: everything earlier tests left behind is given back first
    (torch's cached blocks, the library's table slabs and staging), and on a device with the MI355X's memory (>= 256 GB) the
    full size is ASSERTED, not silently halved (the reference's bar is the whole corpus: src/LZ4.Tests/ConformanceTests.cs:121-133).
    A smaller device still runs the test at the largest size that fits and says so."""
    import gc
    from lz4net_amd import _lib
    gc.collect()
    torch.cuda.synchronize()
    _lib.check(_lib.lib().lz4hip_release_workspaces())
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    n = want
    while n * bytes_per_block * 1.05 > free and n > 1024:
        n //= 2
    if total >= 256 * 10**9:
        assert n == want, f"{free / 2**30:.1f} GiB free of {total / 2**30:.1f}: the full-size test would shrink to {n} blocks"
    elif n != want:
        print(f"full-size test reduced to {n} of {want} blocks: device has {total / 2**30:.1f} GiB")
    return n


@pytest.mark.parametrize("dist", [2, 3], ids=["D2-fuzzer", "D3-records"])
def test_full_size_decode_properties(torch_cuda, oracle, dist):
    """BASELINE config 2 shape (2^20 x 64 KiB, reduced only if the box has less memory): round trip is the
    identity, every result equals the compressed length, checksum of checksums matches the input's, and EVERY block's
    compressed bytes equal the CPU reference's -- for both sequence-dense distributions (D3's long repeats are where the
    encoders' repeat handling is exercised hardest)."""
    torch = torch_cuda
    from lz4net_amd import batch
    n = _full_size(torch, 1 << 20, 2 * batch.BLOCK + batch.BOUND_STRIDE)
    raw = batch.synth(dist, 7, 0, n)
    comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
    back = torch.empty_like(raw)
    used = batch.decode(comp, clen, back, batch.BLOCK)
    assert bool((used == clen).all())
    assert batch.count_mismatches(raw, back, batch.BLOCK) == 0
    a = batch.checksum(raw, batch.BLOCK)
    b = batch.checksum(back, batch.BLOCK)
    assert int(a.sum().item()) == int(b.sum().item()) and bool((a == b).all())
    ratio = float(clen.double().mean().item()) / batch.BLOCK
    if dist == 2:
        assert 0.45 < ratio < 0.53, ratio        # fuzzer-style data compresses to ~0.487
    else:
        assert 0.25 < ratio < 0.40, ratio        # record-like data: ~0.33
    lens = clen.cpu().numpy()
    for i in (0, 1, n // 2, n - 1):
        want = oracle.compress(oracle.gen(dist, 7, i, 1)[0])
        assert lens[i] == len(want) and np.array_equal(comp[i, :len(want)].cpu().numpy(), want), i
    del raw, back
    _compare_whole_corpus(oracle, batch, comp, clen, False, dist, 7)


@pytest.mark.parametrize("n,mix", [(1 << 18, False), (1 << 18, True), (1 << 20, True)], ids=["2^18-homogeneous", "2^18-half-zeros", "2^20-half-zeros"])
def test_lane_decoder_persistent_grid_and_device_side_choice(torch_cuda, n, mix):
    """The lane decoder's second form: a persistent grid whose lanes pull blocks from a counter.  The library takes it for batches of
    one to three residency rounds (2^18 blocks) and, decided ON THE DEVICE by a counting launch, for large batches in which many
    blocks are routed to the wavefront mapping (every second block zeros: the one-block-per-lane form would run its wavefronts
    half empty).  Whatever form runs: every result equals the compressed length, every byte the input; the knob's three settings
    agree."""
    torch = torch_cuda
    from lz4net_amd import batch, _lib
    free, _ = torch.cuda.mem_get_info()
    while n * (2 * batch.BLOCK + batch.BOUND_STRIDE) * 1.05 > free and n > 4096:
        n //= 2
    raw = batch.synth(2, 313, 0, n)
    if mix:
        raw[1::2] = 0
    comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
    back = torch.empty_like(raw)
    for persist in (0, 1, 2):
        with _lib.tuning(decoder_persist=persist):
            back.fill_(0x5A)
            used = batch.decode(comp, clen, back, batch.BLOCK)
            assert bool((used == clen).all()), persist
            assert batch.count_mismatches(raw, back, batch.BLOCK) == 0, persist
            produced = batch.decode(comp, clen, back.fill_(0x5A), batch.BLOCK, known_output_size=False)
            assert bool((produced == batch.BLOCK).all()) and batch.count_mismatches(raw, back, batch.BLOCK) == 0, persist


def test_lane_decoder_persistent_many_restarts(torch_cuda, oracle):
    """The persistent lane decoder with ONE wavefront in its grid ("decoder_groups" = 1): its 64 lanes decode 12 000 small blocks of
    mixed sizes and kinds, ~190 restarts per lane, each restart with whatever the finished block left in flight (an input piece
    requested in its last trips lands AFTER the restart and must be overwritten by the new block's first piece).  EVERY block's
    bytes and result are compared; blocks of 0 .. 3 000 bytes incl. empty ones, sources at odd addresses."""
    torch = torch_cuda
    from lz4net_amd import batch, _lib
    rng = np.random.default_rng(97)
    n = 12000
    sizes = rng.integers(0, 3000, n)
    sizes[::97] = 0
    sizes[1::53] = rng.integers(1, 20, len(sizes[1::53]))
    stride = 3008 + 16
    raw_h = np.zeros((n, stride), np.uint8)
    for k, dist in enumerate((2, 3, 1, 0)):
        rows = oracle.gen(dist, 131 + k, 0, n // 4 + 1, 3000)
        raw_h[k::4, :3000] = rows[:len(raw_h[k::4])]
    for i in range(n):
        raw_h[i, sizes[i]:] = 0
    raw = torch.from_numpy(raw_h).cuda()
    slen = torch.from_numpy(sizes.astype(np.int32)).cuda()
    cstride = 3000 + 3000 // 255 + 16 + 23          # (odd: compressed rows start at odd addresses)
    comp = torch.zeros((n, cstride), dtype=torch.uint8, device="cuda")
    clen = batch.encode(raw, slen, comp, 3000 + 3000 // 255 + 16, src_len_hint=3000)
    assert bool((clen > 0).all())
    back = torch.full((n, stride), 0x77, dtype=torch.uint8, device="cuda")
    with ForcedMapping("LZ4HIP_DECODER", "lane"), _lib.tuning(decoder_persist=1, decoder_groups=1):
        used = batch.decode(comp, clen, back, slen, known_output_size=True)
        torch.cuda.synchronize()
        assert bool((used == clen).all())
        back_h = back.cpu().numpy()
        for i in range(n):
            assert np.array_equal(back_h[i, :sizes[i]], raw_h[i, :sizes[i]]), i
            assert (back_h[i, sizes[i]:] == 0x77).all(), (i, "wrote past the block")
        produced = batch.decode(comp, clen, back.fill_(0x77), slen, known_output_size=False)
        torch.cuda.synchronize()
        assert bool((produced == slen).all())
        assert np.array_equal(back.cpu().numpy(), back_h)


def test_lane_decoder_counter_slots_reused_across_streams(torch_cuda, oracle):
    """The persistent lane decoder's work counters live in a per-device ring of 64 slots; a slot is handed out again after 64
    further launches, on whatever stream.  One long persistent decode on stream A, then 80 short ones spread over eight other
    streams while A is still running: the launch that draws A's slot again must wait for A (the slot's event) instead of zeroing
    a counter A's lanes are still pulling from -- else its own lanes would draw numbers from A's range and decode nothing.
    Every block of every launch is checked."""
    torch = torch_cuda
    from lz4net_amd import batch, _lib
    n_big = 1 << 17
    free, _ = torch.cuda.mem_get_info()
    while n_big * (2 * batch.BLOCK + batch.BOUND_STRIDE) * 1.1 > free and n_big > 4096:
        n_big //= 2
    raw = batch.synth(2, 515, 0, n_big)
    comp = torch.empty((n_big, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
    back = torch.full_like(raw, 0x3C)
    used = torch.zeros(n_big, dtype=torch.int32, device="cuda")
    # the short launches: 256 blocks of 2 KiB each, their own buffers per launch
    m, length, launches = 256, 2048, 80
    small_raw = batch.synth(2, 516, 0, m * launches, length)
    cap = length + length // 255 + 16
    small_comp = torch.zeros((m * launches, cap + 8), dtype=torch.uint8, device="cuda")
    small_len = batch.encode(small_raw, length, small_comp, cap)
    small_back = torch.full_like(small_raw, 0x3C)
    small_used = torch.zeros(m * launches, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(9)]
    with ForcedMapping("LZ4HIP_DECODER", "lane"), _lib.tuning(decoder_persist=1):
        with torch.cuda.stream(streams[0]):
            batch.decode(comp, clen, back, batch.BLOCK, result=used)
        for k in range(launches):
            sl = slice(k * m, (k + 1) * m)
            with torch.cuda.stream(streams[1 + k % 8]):
                batch.decode(small_comp[sl], small_len[sl], small_back[sl], length, result=small_used[sl])
        torch.cuda.synchronize()
    assert bool((used == clen).all()) and batch.count_mismatches(raw, back, batch.BLOCK) == 0
    assert bool((small_used == small_len).all()), "a short launch lost blocks: its counter slot was still in use"
    assert batch.count_mismatches(small_raw, small_back, length) == 0
    _lib.check(_lib.lib().lz4hip_release_workspaces())      # (frees the ring, its events included; the next launch builds a new one)
    with ForcedMapping("LZ4HIP_DECODER", "lane"), _lib.tuning(decoder_persist=1):
        batch.decode(small_comp[:m], small_len[:m], small_back[:m].fill_(0), length, result=small_used[:m].zero_())
    assert bool((small_used[:m] == small_len[:m]).all()) and batch.count_mismatches(small_raw[:m], small_back[:m], length) == 0


def _compare_whole_corpus(oracle, batch, comp, clen, hc, dist, seed, budget=240.0):
    """EVERY block's (compressed length, checksum of the compressed bytes) from the GPU rows against the CPU codec, which
    regenerates the block from its seed, compresses it and keeps only those two numbers (oracle/batch.c
    lz4o_verify_stream, all host cores) -- the reference's own bar is identity over the whole corpus
    (src/LZ4.Tests/ConformanceTests.cs:121-133).  First / last 64 blocks and every 4096-th are compared byte for byte
    as well."""
    import os
    from oracle.oracle import Reference
    codec = Reference() if Reference.available() else oracle
    n = comp.shape[0]
    g_sum = batch.checksum(comp, clen).cpu().numpy().view(np.uint64)
    g_len = clen.cpu().numpy()
    done, c_len, c_sum = oracle.verify_stream(codec, hc, dist, seed, 0, 1, n, threads=os.cpu_count() or 1, budget_seconds=budget)
    assert done >= min(n, 4096), ("the CPU side got through too few blocks to call this a corpus check", done)
    bad = np.nonzero((g_len[:done] != c_len[:done]) | (g_sum[:done] != c_sum[:done]))[0]
    assert bad.size == 0, (hc, "blocks differing from the CPU codec", bad[:8], "of", done)
    print(f"whole-corpus check hc={hc}: {done} of {n} blocks identical to the CPU {codec.kind}")
    for i in sorted(set(list(range(0, min(64, n))) + list(range(max(n - 64, 0), n)) + list(range(0, n, 4096)))):
        want = codec.compress(oracle.gen(dist, seed, i, 1)[0], hc=hc)
        assert g_len[i] == len(want) and np.array_equal(comp[i, :len(want)].cpu().numpy(), want), (hc, i)


@pytest.mark.parametrize("dist", [2, 3], ids=["D2-fuzzer", "D3-records"])
def test_full_size_hc_encode_whole_corpus(torch_cuda, oracle, dist):
    """BASELINE configs[3] shape (2^18 x 64 KiB, LZ4HC): every block's compressed bytes equal the CPU codec's
    (length + checksum for all of them, bytes for the sample), and the batch round-trips -- D2 and D3."""
    torch = torch_cuda
    from lz4net_amd import batch
    n = _full_size(torch, 1 << 18, 2 * batch.BLOCK + batch.BOUND_STRIDE + 200000)
    raw = batch.synth(dist, 11, 0, n)
    comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND, hc=True)
    assert bool((clen > 0).all())
    back = torch.empty_like(raw)
    used = batch.decode(comp, clen, back, batch.BLOCK)
    assert bool((used == clen).all()) and batch.count_mismatches(raw, back, batch.BLOCK) == 0
    del back
    _compare_whole_corpus(oracle, batch, comp, clen, True, dist, 11)


# ---- limited output through the DEVICE-pointer entry point: the kernel's own writes against guard bytes -------------
_LIMITED_CASES = [("fast", "LZ4HIP_ENCODER", "wave"), ("fast", "LZ4HIP_ENCODER", "lane"), ("hc", "LZ4HIP_HC", "wave"), ("hc", "LZ4HIP_HC", "lane")]


@pytest.mark.parametrize("mode,var,mapping", _LIMITED_CASES, ids=[f"{m}-{w}" for m, _, w in _LIMITED_CASES])
def test_device_limited_output_guard_bytes(torch_cuda, oracle, mode, var, mapping):
    """original/fuzzer.c:212-224 against the KERNELS: lz4hip_encode_batch_device with per-block capacities of exactly the
    compressed size (must succeed, identical bytes), one byte less and seven bytes less (must return 0), the destination
    rows sitting inside one device tensor pre-filled with 0xA5.  No byte at or past a row's capacity may change, whatever
    the result -- the host-pointer tests cannot see that (their scatter clamps to the capacity), this one reads the
    kernel's own buffer back.  Blocks of every distribution and of sizes around the kernels' internal limits."""
    torch = torch_cuda
    from lz4net_amd import batch
    hc = mode == "hc"
    sizes = (13, 14, 300, 4096, 20000, 65535, 65536)
    blocks = []
    for dist in range(4):
        for k, length in enumerate(sizes):
            blocks.append(oracle.gen(dist, 90 + k, dist * 100 + k, 1, length)[0][:length])
    if mapping == "lane":
        # the lane mappings take whole batches: many copies so that every lane of a few wavefronts has work
        blocks = blocks * 8
    want = [oracle.compress(a, hc=hc) for a in blocks[:len(sizes) * 4]]
    want = want * (len(blocks) // len(want))
    n = len(blocks)
    src_stride = 65536 + 32
    src = torch.zeros((n, src_stride), dtype=torch.uint8, device="cuda")
    for i, a in enumerate(blocks):
        src[i, :len(a)] = torch.from_numpy(a).cuda()
    src_len = torch.tensor([len(a) for a in blocks], dtype=torch.int32, device="cuda")
    row = batch.BOUND_STRIDE + 64
    for delta in (0, 1, 7):
        caps = [max(len(w) - delta, 0) for w in want]
        dst = torch.full((n, row), 0xA5, dtype=torch.uint8, device="cuda")
        cap_t = torch.tensor(caps, dtype=torch.int32, device="cuda")
        with ForcedMapping(var, mapping):
            res = batch.encode(src, src_len, dst, cap_t, hc=hc, src_len_hint=65536)
            torch.cuda.synchronize()
        res_h, dst_h = res.cpu().numpy(), dst.cpu().numpy()
        for i, w in enumerate(want):
            if delta == 0:
                assert res_h[i] == len(w), (mode, mapping, i, len(blocks[i]), res_h[i], len(w))
                assert np.array_equal(dst_h[i, :len(w)], w), (mode, mapping, i)
            else:
                assert res_h[i] == 0, (mode, mapping, delta, i, len(blocks[i]), res_h[i], len(w))
            assert (dst_h[i, caps[i]:] == 0xA5).all(), (mode, mapping, delta, i, len(blocks[i]), "the kernel wrote past the capacity",
                                                        int(np.nonzero(dst_h[i, caps[i]:] != 0xA5)[0][0]) + caps[i])


def test_round_robin_sharding_single_process(torch_cuda, oracle):
    # world_size-1 view of the N>1 path: the shard helper must reproduce global order
    torch = torch_cuda
    from lz4net_amd import batch
    n, world = 37, 4
    parts = []
    for r in range(world):
        cnt = batch.local_block_count(n, r, world)
        idx = [batch.local_to_global(j, r, world) for j in range(cnt)]
        assert all(i % world == r and i < n for i in idx)
        parts.append(idx)
    assert sorted(sum(parts, [])) == list(range(n))


def test_concurrent_streams_share_the_workspace_safely(torch_cuda, oracle):
    """The lane encoders keep their tables in a per-device workspace; launches from different host threads / streams
    must not overlap on it (lz4hip_api.hip: workspace leases).  Three threads, each on its own stream, encode
    different batches that are large enough for the lane mapping; every result must be the oracle's."""
    import threading
    torch = torch_cuda
    from lz4net_amd import batch
    n, length = 49152, 4096                                          # (>= 49152 blocks: the default dispatch launches the lane mapping too)
    bound = length + length // 255 + 16
    errors = []

    def worker(t):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for rep in range(3):
                    raw = batch.synth(2, 100 + 10 * t + rep, 0, n, length=length)
                    comp = torch.empty((n, bound + 15), dtype=torch.uint8, device="cuda")
                    clen = batch.encode(raw, length, comp, bound)
                    s.synchronize()
                    lens = clen.cpu().numpy()
                    for i in (0, 1, 777, n // 2, n - 1):
                        want = oracle.compress(oracle.gen(2, 100 + 10 * t + rep, i, 1, length=length)[0])
                        got = comp[i, :lens[i]].cpu().numpy()
                        if lens[i] != len(want) or not np.array_equal(got, want):
                            errors.append((t, rep, i))
        except Exception as e:                                       # surfaces in the main thread
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_lane_encoder_many_blocks_per_lane(torch_cuda, oracle):
    """Lane encoder with one wavefront per CU and 2^20 small blocks: every lane encodes ~64 blocks in a row, so its
    epoch-stamped table wraps at least once; sampled blocks must be the oracle's bytes and all must round-trip."""
    torch = torch_cuda
    from lz4net_amd import batch, _lib
    n, length = 1 << 20, 256
    bound = length + length // 255 + 16
    raw = batch.synth(2, 31, 0, n, length=length)
    comp = torch.empty((n, bound + 15), dtype=torch.uint8, device="cuda")
    with ForcedMapping("LZ4HIP_ENCODER", "lane"), _lib.tuning(encoder_waves_per_cu=1):
        clen = batch.encode(raw, length, comp, bound)
    back = torch.empty_like(raw)
    used = batch.decode(comp, clen, back, length)
    assert bool((clen > 0).all()) and bool((used == clen).all())
    assert batch.count_mismatches(raw, back, length) == 0
    lens = clen.cpu().numpy()
    for i in list(range(0, n, 65521)) + [n - 1]:
        want = oracle.compress(oracle.gen(2, 31, i, 1, length=length)[0])
        assert lens[i] == len(want) and np.array_equal(comp[i, :lens[i]].cpu().numpy(), want), i


def test_lane_encoder_slab_chunks_measured_and_rebuilt(torch_cuda, oracle):
    """The lane encoder's table slab is a set of separately allocated chunks whose placement the library measures.  A batch
    that needs the full residency (16 wavefronts per CU: 16 or 64 chunks) must give the oracle's bytes; the slab reports its measured rate and how
    many candidates were built; lz4hip_release_workspaces gives the chunks back (free memory returns) and the next call builds a slab again --
    this time the first candidate, unmeasured (knob encoder_slab_tries = 1) -- with the same bytes."""
    torch = torch_cuda
    from lz4net_amd import batch, _lib
    n, length = 1 << 18, 1024                                        # 2^18 blocks = one per lane of the full grid
    bound = length + length // 255 + 16
    raw = batch.synth(2, 77, 0, n, length=length)
    comp = torch.empty((n, bound + 15), dtype=torch.uint8, device="cuda")
    _lib.check(_lib.lib().lz4hip_release_workspaces())
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    with ForcedMapping("LZ4HIP_ENCODER", "lane"):
        clen = batch.encode(raw, length, comp, bound)
        torch.cuda.synchronize()
        held = free0 - torch.cuda.mem_get_info()[0]
        assert held >= 8 << 30, "the slab of a full grid is 8 GiB of tables"
        tried, rate = _lib.tuning_get("encoder_slab_tried"), _lib.tuning_get("encoder_slab_rate")
        assert 1 <= tried <= 4 and rate > 5000, (tried, rate)        # measured: G steps per second x 1000 (20 000 .. 27 000 on an MI355X)
        first = comp.clone()
        _lib.check(_lib.lib().lz4hip_release_workspaces())
        torch.cuda.synchronize()
        assert free0 - torch.cuda.mem_get_info()[0] < 1 << 30, "the chunks were not given back"
        assert _lib.tuning_get("encoder_slab_tried") == 0
        comp.zero_()
        with _lib.tuning(encoder_slab_tries=1):
            clen2 = batch.encode(raw, length, comp, bound)
        torch.cuda.synchronize()
        assert _lib.tuning_get("encoder_slab_tried") == 1 and _lib.tuning_get("encoder_slab_rate") == 0
    assert bool((clen == clen2).all())
    lens = clen.cpu().numpy()
    assert bool((clen > 0).all())
    width = int(lens.max())
    assert bool(torch.equal(first[:, :width] * (torch.arange(width, device="cuda")[None, :] < clen[:, None]),
                            comp[:, :width] * (torch.arange(width, device="cuda")[None, :] < clen2[:, None])))
    for i in list(range(0, n, 32749)) + [n - 1]:
        want = oracle.compress(oracle.gen(2, 77, i, 1, length=length)[0])
        assert lens[i] == len(want) and np.array_equal(comp[i, :lens[i]].cpu().numpy(), want), i
    back = torch.empty_like(raw)
    used = batch.decode(comp, clen2, back, length)
    assert bool((used == clen2).all()) and batch.count_mismatches(raw, back, length) == 0


def test_lane_encoder_slab_under_memory_pressure(torch_cuda, oracle):
    """With ~5 GiB of device memory free the 8 GiB slab of the full residency cannot be had: a candidate fails half way through its chunks (they are
    given back), the launch halves the residency until a slab fits, and the bytes are still the oracle's."""
    torch = torch_cuda
    from lz4net_amd import batch, _lib
    n, length = 1 << 18, 1024
    bound = length + length // 255 + 16
    raw = batch.synth(2, 78, 0, n, length=length)
    comp = torch.empty((n, bound + 15), dtype=torch.uint8, device="cuda")
    back = torch.empty_like(raw)
    _lib.check(_lib.lib().lz4hip_release_workspaces())
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    hog = torch.empty(free0 - (5 << 30), dtype=torch.uint8, device="cuda")
    try:
        free1 = torch.cuda.mem_get_info()[0]
        assert free1 < 6 << 30
        with ForcedMapping("LZ4HIP_ENCODER", "lane"):
            clen = batch.encode(raw, length, comp, bound)
        torch.cuda.synchronize()
        held = free1 - torch.cuda.mem_get_info()[0]
        assert (1 << 30) <= held <= free1, held                      # a slab of a lower residency (4 GiB or 2 GiB), not the 8 GiB one
        used = batch.decode(comp, clen, back, length)
        assert bool((clen > 0).all()) and bool((used == clen).all()) and batch.count_mismatches(raw, back, length) == 0
        lens = clen.cpu().numpy()
        for i in list(range(0, n, 32719)) + [n - 1]:
            want = oracle.compress(oracle.gen(2, 78, i, 1, length=length)[0])
            assert lens[i] == len(want) and np.array_equal(comp[i, :lens[i]].cpu().numpy(), want), i
    finally:
        del hog
        _lib.check(_lib.lib().lz4hip_release_workspaces())
        torch.cuda.empty_cache()


def test_fast_encode_default_dispatch_two_launches(torch_cuda, oracle):
    """Default dispatch of a large fast-encode batch: the wavefront mapping runs over every block and hands the blocks made
    of short sequences over to the lane mapping.  A batch mixing incompressible, fuzzer-style, record-like and zero blocks
    must launch BOTH kernels, leave no block with the hand-over marker, and produce the reference's bytes for every block
    sampled, whichever kernel finished it."""
    torch = torch_cuda
    from lz4net_amd import _lib, batch
    per = 12288                                                      # (4 x 12288 = 49152 blocks: from there on the default dispatch launches both mappings)
    parts = [batch.synth(d, 4242, 0, per) for d in (1, 2, 3, 0)]
    raw = torch.cat(parts, dim=0)
    n = raw.shape[0]
    comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    before = _lib.dispatch_counts()
    clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
    after = _lib.dispatch_counts()
    assert after[2] > before[2] and after[3] > before[3], "both fast-encode mappings must have been launched"
    assert bool((clen > 0).all()), "a block was left with the hand-over marker (or failed)"
    back = torch.empty_like(raw)
    used = batch.decode(comp, clen, back, batch.BLOCK)
    assert bool((used == clen).all()) and batch.count_mismatches(raw, back, batch.BLOCK) == 0
    lens = clen.cpu().numpy()
    for k, d in enumerate((1, 2, 3, 0)):
        for j in (0, 1, per // 2, per - 1):
            i = k * per + j
            want = oracle.compress(oracle.gen(d, 4242, j, 1)[0])
            assert lens[i] == len(want) and np.array_equal(comp[i, :lens[i]].cpu().numpy(), want), (d, j)


@pytest.mark.parametrize("decoder", ["wave", "lane"])
def test_decode_into_unaligned_rows(torch_cuda, oracle, decoder):
    """Destination rows at odd addresses and an odd stride, sources at odd addresses too: the 16-byte stores of both
    decoder mappings (cooperative 64-byte flush of the lane mapping, register fills of the wavefront mapping) must
    not depend on alignment, and must not touch the bytes between the rows."""
    torch = torch_cuda
    from lz4net_amd import batch
    n = 2048
    for dist in (0, 2, 3):
        raw = batch.synth(dist, 5, 0, n)
        comp0 = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
        clen = batch.encode(raw, batch.BLOCK, comp0, batch.BOUND)
        sstride = batch.BOUND_STRIDE + 5
        sbuf = torch.zeros(n * sstride + 64, dtype=torch.uint8, device="cuda")
        comp = sbuf[7:7 + n * sstride].view(n, sstride)
        comp[:, :batch.BOUND_STRIDE] = comp0
        dstride = batch.BLOCK + 13
        dbuf = torch.full((n * dstride + 64,), 0xC3, dtype=torch.uint8, device="cuda")
        back = dbuf[3:3 + n * dstride].view(n, dstride)
        with ForcedMapping("LZ4HIP_DECODER", decoder):
            used = batch.decode(comp, clen, back, batch.BLOCK)
            torch.cuda.synchronize()
        assert bool((used == clen).all())
        assert bool((back[:, :batch.BLOCK] == raw).all()), (decoder, dist)
        assert bool((back[:, batch.BLOCK:] == 0xC3).all()) and bool((dbuf[:3] == 0xC3).all())


@pytest.mark.parametrize("dist", [2, 3])
def test_unknown_size_decode_at_scale(torch_cuda, dist):
    """LZ4_uncompress_unknownOutputSize semantics (Decode(..., knownOutputLength: false), the reference's default)
    through the lane mapping on a batch large enough to select it: the result is the number of bytes produced."""
    torch = torch_cuda
    from lz4net_amd import batch
    n = 1 << 16
    raw = batch.synth(dist, 77, 0, n)
    comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
    stride = batch.BLOCK + 64
    back = torch.full((n, stride), 0x3C, dtype=torch.uint8, device="cuda")
    produced = batch.decode(comp, clen, back, batch.BLOCK + 40, known_output_size=False)
    assert bool((produced == batch.BLOCK).all())
    assert bool((back[:, :batch.BLOCK] == raw).all())
    assert bool((back[:, batch.BLOCK:] == 0x3C).all())
    # a capacity that is too small is an error at some position of the source: negative result, nothing past the capacity
    back.fill_(0x3C)
    short = batch.decode(comp, clen, back, batch.BLOCK - 1, known_output_size=False)
    assert bool((short < 0).all())
    assert bool((back[:, batch.BLOCK - 1:] == 0x3C).all())
