"""Runs the REAL kernel sources (lz4net_amd/csrc/*.hpp) under the CPU SIMT emulator (tests/simt/) and
checks them against the oracle: bit-exact compressed bytes, bit-exact decoded bytes, identical return
codes.  This is how kernel logic is validated in the GPU-less build container; the `-m gpu` tests repeat
the same comparisons through the C-ABI on a real MI355X.  Nothing here is a product code path."""
import numpy as np
import pytest

import emu_helpers as emu
from oracle.oracle import compress_bound

SIZES = (0, 1, 12, 13, 14, 64, 65, 300, 4096, 20000, 65535, 65536)


def _mapping(m):
    if isinstance(m, str) and m.startswith("l4w"):                   # "l4w1" / "l4w2": the default configuration in workgroups of four wavefronts, dual ring stores / wrapped rows
        return dict(lane=int(m[3:]), gen=6)
    if isinstance(m, str) and m.startswith("l4p"):                   # "l4p2": persistent variant of the default generation-4 configuration, 2 wavefronts
        return dict(lane=int(m[3:]), gen=5)
    if isinstance(m, str) and m.startswith("l4c"):                   # "l4c1192": fourth-generation lane decoder, ring 192 + 1000 x variant
        return dict(lane=int(m[3:]), gen=4)
    if isinstance(m, str) and m.startswith("l3r"):                   # "l3r240": third-generation lane decoder, ring 240 (staging 64)
        ring, _, stage = m[3:].partition("s")
        return dict(lane=int(ring), stage=int(stage or 64), gen=3)
    if isinstance(m, str):
        ring, _, stage = m[4:].partition("s")                        # "lane128s64": ring 128, staging 64
        return dict(lane=int(ring), stage=int(stage or 64))
    return {}


LANE3 = ["l3r128", "l3r240", "l4c27192", "l4c59192", "l4p1", "l4p3", "l4w1", "l4w2"]    # third generation (tools/ab, A/B only): a power-of-two ring and another one; the product's lane decoder: the round-4 default, the default (sector input), its persistent form with one / three wavefronts
LANE3_ALL = ["l3r128", "l3r240", "l4c128", "l4c192", "l4c1192", "l4c7192", "l4c2240", "l4c5256", "l4c27192", "l4c35192", "l4c59192"]


def _blocks(oracle, sizes=SIZES, seeds=(5,)):
    out = []
    for dist in range(4):
        for seed in seeds:
            for n in sizes:
                out.append(oracle.gen(dist, seed, n, 1, max(n, 1))[0][:n])
    rng = np.random.default_rng(7)
    for n in (50, 700, 9000):
        for k in (2, 3, 16):
            out.append(rng.integers(0, k, n, dtype=np.uint8))
    out.append(np.frombuffer(b"abcabcabcabcabcabcabcabcabcabc" * 40, dtype=np.uint8))
    return out


@pytest.mark.parametrize("lane", [False, "lane128s64", "lane256s128"] + LANE3_ALL, ids=["wave-per-block", "lane128s64", "lane256s128"] + LANE3_ALL)
def test_decode_known_size(oracle, lane):
    blocks = _blocks(oracle)
    for hc in (False, True):
        comps = [oracle.compress(a, hc=hc) for a in blocks]
        res, dst = emu.decode(comps, [a.size for a in blocks], known=True, waves_per_group=2, **_mapping(lane))
        for i, (a, c) in enumerate(zip(blocks, comps)):
            assert res[i] == len(c), (i, hc, res[i], len(c))
            assert np.array_equal(dst[i, :a.size], a), (i, hc)
            assert (dst[i, a.size:] == 0xA5).all(), (i, hc, "wrote past the block")


def test_decode_partitioned_between_mappings(oracle):
    # default dispatch: every block is decoded by exactly one of the two mappings
    blocks = _blocks(oracle, sizes=(13, 300, 4096, 65536))
    comps = [oracle.compress(a) for a in blocks]
    res, dst = emu.decode(comps, [a.size for a in blocks], known=True, auto=True)
    for i, (a, c) in enumerate(zip(blocks, comps)):
        assert res[i] == len(c) and np.array_equal(dst[i, :a.size], a), i


@pytest.mark.parametrize("lane", [False] + LANE3, ids=["wave-per-block"] + LANE3)
def test_decode_unknown_size(oracle, lane):
    blocks = _blocks(oracle, sizes=(0, 1, 13, 300, 4096, 65536))
    comps = [oracle.compress(a) for a in blocks]
    for extra in (0, 1, 100):
        res, dst = emu.decode(comps, [a.size + extra for a in blocks], known=False, **_mapping(lane))
        for i, (a, c) in enumerate(zip(blocks, comps)):
            assert res[i] == a.size, (i, extra, res[i])
            assert np.array_equal(dst[i, :a.size], a)
            assert (dst[i, a.size + extra:] == 0xA5).all()


@pytest.mark.parametrize("lane", [False] + LANE3, ids=["wave-per-block"] + LANE3)
def test_decode_error_codes_match_oracle(oracle, lane):
    # wrong sizes and corrupted streams: same (negative) return codes as the reference decoders
    rng = np.random.default_rng(11)
    blocks = _blocks(oracle, sizes=(13, 300, 4096, 65536))
    comps = [oracle.compress(a) for a in blocks]
    cases_k, want_k, cases_u, want_u = [], [], [], []
    for a, c in zip(blocks, comps):
        for osize in (a.size - 1, a.size + 1, a.size // 2):
            cases_k.append((c, osize)); want_k.append(oracle.uncompress_raw(c, osize)[0])
        for isz, mo in ((len(c), a.size - 1), (len(c) - 1, a.size), (len(c) + 1, a.size), (0, a.size)):
            cases_u.append((c, isz, mo)); want_u.append(oracle.uncompress_unknown_raw(c, isz, mo)[0])
        for _ in range(4):
            cc = c.copy()
            cc[rng.integers(0, len(c))] = rng.integers(0, 256)
            cases_k.append((cc, a.size)); want_k.append(oracle.uncompress_raw(cc, a.size)[0])
            cases_u.append((cc, len(cc), a.size)); want_u.append(oracle.uncompress_unknown_raw(cc, len(cc), a.size)[0])
    # known-size: give the kernel a generous source length (zero padded) like the oracle wrapper does
    pad = [np.concatenate([c, np.zeros(max(o, 0) + 1024, np.uint8)]) for c, o in cases_k]
    res, dst = emu.decode(pad, [o for _, o in cases_k], known=True, **_mapping(lane))
    for i, w in enumerate(want_k):
        assert res[i] == w, ("known", i, res[i], w)
        assert (dst[i, max(cases_k[i][1], 0):] == 0xA5).all()
    padu = [np.concatenate([c, np.zeros(8, np.uint8)]) for c, _, _ in cases_u]
    res, dst = emu.decode(padu, [mo for _, _, mo in cases_u], known=False, src_lens=[i for _, i, _ in cases_u], **_mapping(lane))
    for i, w in enumerate(want_u):
        assert res[i] == w, ("unknown", i, res[i], w)
        assert (dst[i, max(cases_u[i][2], 0):] == 0xA5).all()


@pytest.mark.parametrize("lane", [False, True, "wg5"], ids=["wave-per-block", "lane-per-block", "five-blocks-per-workgroup"])
def test_encode_fast_bit_exact(oracle, lane):
    blocks = _blocks(oracle, sizes=SIZES + (65546, 65547, 70000), seeds=(5, 6) if lane is True else (5,))
    if lane == "wg5":
        blocks = blocks[:len(blocks) - (len(blocks) % 5 == 0)]       # a last workgroup with idle wavefronts
    res, dst = emu.encode(blocks, lane=lane is True, wg5=lane == "wg5")
    for i, a in enumerate(blocks):
        want = oracle.compress(a)
        assert res[i] == len(want), (i, a.size, res[i], len(want))
        assert np.array_equal(dst[i, :res[i]], want), (i, a.size)
        assert (dst[i, compress_bound(a.size):] == 0xA5).all()


@pytest.mark.parametrize("lane", [False, True], ids=["wave-per-block", "lane-per-block"])
def test_encode_fast_limited_output(oracle, lane):
    # original/fuzzer.c:212-227: exact capacity succeeds, one byte less returns 0, canary untouched
    blocks = _blocks(oracle, sizes=(13, 300, 4096, 65536))
    lens = [len(oracle.compress(a)) for a in blocks]
    for delta in (0, -1, -7):
        caps = [max(l + delta, 0) for l in lens]
        res, dst = emu.encode(blocks, caps=caps, lane=lane)
        for i, a in enumerate(blocks):
            want = oracle.compress_raw(a, caps[i])[0]
            assert res[i] == want, (i, delta, res[i], want)
            assert (dst[i, caps[i]:] == 0xA5).all(), (i, delta, "wrote past the capacity")


def test_encode_fast64k_window_wide_count_and_single_store_paths(oracle):
    """The second version of the 64k fast encoder (lz4hip_encode.hpp, encode_fast_block64k): its register window (re-based every ~244 bytes
    and after every long match), the combined candidate test + 256-byte count (and its continuation past 256 equal bytes, and the byte-wise
    form within 261 bytes of the end), the one-store sequence emit (<= 14 literals, match < 19) next to the general one -- on sizes around
    every threshold, dense / sparse / periodic / long-run data, with exact and too-small output limits (return value and guard bytes)."""
    rng = np.random.default_rng(606)
    blocks = []
    for n in list(range(13, 40)) + [255, 256, 257, 260, 261, 262, 263, 268, 269, 270, 300, 511, 512, 513, 517, 518, 519, 777, 1200, 5000, 33333, 65536, 65546]:
        for dist in (2, 3):
            blocks.append(oracle.gen(dist, 600 + n, n, 1, n)[0][:n])
    for n in (300, 1000, 4000, 20000):
        blocks.append(rng.integers(0, 2, n, dtype=np.uint8))                                   # long runs of near-matches
        blocks.append(np.tile(rng.integers(0, 256, 7, dtype=np.uint8), n // 7 + 1)[:n])       # period 7: one match of n - 12 bytes
        a = np.tile(rng.integers(0, 256, 300, dtype=np.uint8), n // 300 + 1)[:n].copy()       # period 300 with a few damaged bytes: matches > 256
        a[rng.integers(0, n, max(n // 900, 1))] ^= 0x55
        blocks.append(a)
        b = rng.integers(0, 256, n, dtype=np.uint8)                                            # incompressible with islands of copies
        for _ in range(n // 200):
            src, ln, dstp = int(rng.integers(0, n - 40)), int(rng.integers(4, 40)), int(rng.integers(40, n - 40))
            b[dstp:dstp + ln] = b[src:src + ln]
        blocks.append(b)
    want = [oracle.compress(a) for a in blocks]
    res, dst = emu.encode(blocks)
    for i, (a, w) in enumerate(zip(blocks, want)):
        assert res[i] == len(w), (i, a.size, res[i], len(w))
        assert np.array_equal(dst[i, :res[i]], w), (i, a.size)
        assert (dst[i, compress_bound(a.size):] == 0xA5).all()
    for delta in (0, -1, -3, -9):
        caps = [max(len(w) + delta, 0) for w in want]
        res, dst = emu.encode(blocks, caps=caps)
        for i, a in enumerate(blocks):
            r = oracle.compress_raw(a, caps[i])[0]
            assert res[i] == r, (i, a.size, delta, res[i], r)
            assert (dst[i, caps[i]:] == 0xA5).all(), (i, delta, "wrote past the capacity")
            if r > 0:
                assert np.array_equal(dst[i, :r], want[i]), (i, delta)


def test_encode_fast_two_launches_hand_over(oracle):
    """Default dispatch of a large batch: the wavefront mapping hands blocks made of short sequences over to the lane
    mapping (kDeferredResult) and finishes the others; every block ends up with the reference's bytes whoever encoded it."""
    blocks = _blocks(oracle, sizes=(0, 13, 300, 4096, 20000, 65536, 65547))
    half = oracle.gen(1, 9, 0, 1, 65536)[0].copy()          # incompressible start, dense matches later: handed over late
    half[40000:] = oracle.gen(2, 9, 0, 1, 65536)[0][:25536]
    blocks.append(half)
    res, dst, deferred = emu.encode_two_launches(blocks)
    for i, a in enumerate(blocks):
        want = oracle.compress(a)
        assert res[i] == len(want), (i, a.size, res[i], len(want))
        assert np.array_equal(dst[i, :res[i]], want), (i, a.size)
    # incompressible blocks stay with the wavefront mapping, fuzzer-style blocks of some length are handed over
    kinds = {}
    for i, a in enumerate(blocks):
        kinds.setdefault(bool(deferred[i]), []).append(a.size)
    assert True in kinds and False in kinds, kinds
    assert deferred[-1] == 1
    # limited output through both launches
    lens = [len(oracle.compress(a)) for a in blocks]
    for delta in (0, -1):
        caps = [max(l + delta, 0) for l in lens]
        res, dst, _ = emu.encode_two_launches(blocks, caps=caps)
        for i, a in enumerate(blocks):
            assert res[i] == oracle.compress_raw(a, caps[i])[0], (i, delta)
            assert (dst[i, caps[i]:] == 0xA5).all(), (i, delta, "wrote past the capacity")


def test_synth_generators_match_cpu_twins(oracle):
    for dist in range(4):
        for length in (1, 100, 4096, 65536):
            got = emu.synth(dist, 77, 1000, 3, length)
            want = oracle.gen(dist, 77, 1000, 3, length)
            assert np.array_equal(got[:, :length], want[:, :length]), (dist, length)
    # a rank's round-robin share: row i is block first + i * step
    got = emu.synth(2, 5, 3, 4, 300, block_step=8)
    for i in range(4):
        assert np.array_equal(got[i], oracle.gen(2, 5, 3 + 8 * i, 1, 300)[0])


def test_checksum_and_compare(oracle):
    rows = [oracle.gen(2, 1, i, 1, n)[0][:n] for i, n in enumerate((0, 1, 7, 8, 9, 1000, 65536))]
    sums = emu.checksum(rows)
    for r, s in zip(rows, sums):
        assert int(s) == oracle.checksum(r)
    a = oracle.gen(1, 3, 0, 5, 5000)
    b = a.copy()
    b[1, 17] ^= 1; b[4, 4999] ^= 0x80; b[4, 4998] ^= 0x80
    assert emu.compare(a, a, [5000] * 5) == 0
    assert emu.compare(a, b, [5000] * 5) == 3


@pytest.mark.parametrize("lane", [False, True, "conv"], ids=["wave-per-block", "lane-per-block", "lane-per-block-convergent"])
def test_encode_hc_bit_exact(oracle, lane):
    # LZ4HC: several blocks per persistent workgroup (stale-state check), 16- and 32-bit heads
    sizes = (0, 1, 12, 13, 14, 64, 300, 4096) if not lane else (0, 1, 12, 13, 14, 64, 300, 4096, 20000, 65536)
    blocks = _blocks(oracle, sizes=sizes)
    conv = lane == "conv"
    res, dst = emu.encode(blocks, hc=True, groups=2, lane=bool(lane), conv=conv)
    for i, a in enumerate(blocks):
        want = oracle.compress(a, hc=True)
        assert res[i] == len(want), (i, a.size, res[i], len(want))
        assert np.array_equal(dst[i, :res[i]], want), (i, a.size)
    big = [oracle.gen(2, 3, 0, 1, 70000)[0]] + ([oracle.gen(3, 3, 0, 1, 65536)[0]] if not lane else [])
    res, dst = emu.encode(big, hc=True, groups=1, lane=bool(lane), conv=conv)
    for i, a in enumerate(big):
        want = oracle.compress(a, hc=True)
        assert res[i] == len(want) and np.array_equal(dst[i, :res[i]], want), (i, a.size)


@pytest.mark.parametrize("lane", [False, True, "conv"], ids=["wave-per-block", "lane-per-block", "lane-per-block-convergent"])
def test_encode_hc_limited_output(oracle, lane):
    blocks = _blocks(oracle, sizes=(13, 300, 4096) if lane else (13, 300))
    lens = [len(oracle.compress(a, hc=True)) for a in blocks]
    for delta in (0, -1, -7):
        caps = [max(l + delta, 0) for l in lens]
        res, dst = emu.encode(blocks, caps=caps, hc=True, lane=bool(lane), conv=lane == "conv")
        for i, a in enumerate(blocks):
            want = oracle.compress_raw(a, caps[i], hc=True)[0]
            assert res[i] == want, (i, delta, res[i], want)
            assert (dst[i, caps[i]:] == 0xA5).all(), (i, delta, "wrote past the capacity")


PRECOMPUTED = pytest.mark.parametrize("kind", ["nat", "lcp"], ids=["natural-chains", "chains-with-shared-lengths"])


@PRECOMPUTED
def test_encode_hc_nat_bit_exact(oracle, kind):
    """LZ4HC over precomputed tables: lz4hip_hc_nat.hpp (chain builder + lane kernel without the insert loop) and
    lz4hip_hc_lcp.hpp (the chains also carry the length each position shares with its predecessor: walks read no input)."""
    blocks = _blocks(oracle, sizes=(0, 1, 4, 5, 12, 13, 14, 64, 65, 66, 300, 4096, 20000, 65536))
    blocks.append(oracle.gen(3, 3, 0, 1, 65536)[0])
    blocks.append(np.zeros(65536, np.uint8))
    blocks.append(np.tile(np.frombuffer(b"abc", np.uint8), 21846)[:65536].copy())
    res, dst = emu.encode(blocks, hc=True, groups=2, **{kind: True})
    for i, a in enumerate(blocks):
        want = oracle.compress(a, hc=True)
        assert res[i] == len(want), (i, a.size, res[i], len(want))
        assert np.array_equal(dst[i, :res[i]], want), (i, a.size)
    # a block > 64 KiB is refused by these kernels (the library launches lz4hip_hc_conv.hpp for such batches)
    res, dst = emu.encode([oracle.gen(2, 3, 0, 2).reshape(-1)[:70000].copy(), blocks[10]], hc=True, groups=1, **{kind: True})
    assert res[0] == -2000000002 and res[1] == len(oracle.compress(blocks[10], hc=True))


@PRECOMPUTED
def test_encode_hc_nat_limited_output(oracle, kind):
    blocks = _blocks(oracle, sizes=(13, 300, 4096))
    lens = [len(oracle.compress(a, hc=True)) for a in blocks]
    for delta in (0, -1, -7):
        caps = [max(l + delta, 0) for l in lens]
        res, dst = emu.encode(blocks, caps=caps, hc=True, groups=1, **{kind: True})
        for i, a in enumerate(blocks):
            want = oracle.compress_raw(a, caps[i], hc=True)[0]
            assert res[i] == want, (i, delta, res[i], want)
            assert (dst[i, caps[i]:] == 0xA5).all(), (i, delta, "wrote past the capacity")


@PRECOMPUTED
def test_encode_hc_nat_repeats_and_collisions(oracle, kind):
    """What the exactness arguments of lz4hip_hc_nat.hpp / lz4hip_hc_lcp.hpp rest on: runs of period 1-5 (the repeat optimisation rewires their
    chains and skips their head updates), tiny alphabets (same words everywhere), copies of earlier content, runs with single
    disturbed bytes -- ONE wavefront, two blocks per lane, every block the oracle's bytes."""
    rng = np.random.default_rng(23)
    blocks = []
    for i in range(128):
        mode = i % 5
        sz = 65536 - int(rng.integers(0, 3000)) if i in (2, 68) else int(rng.integers(13, 3000))
        if mode == 0:
            row = rng.integers(0, int(rng.integers(2, 4)), sz).astype(np.uint8)
        elif mode == 1:
            row = rng.integers(0, 256, sz).astype(np.uint8)
            pos = 0
            while pos < sz:
                per, ln = int(rng.integers(1, 6)), int(rng.integers(4, 300))
                seg = np.tile(rng.integers(0, 3, per).astype(np.uint8), ln // per + 2)[:ln]
                e = min(sz, pos + ln)
                row[pos:e] = seg[:e - pos]
                pos = e + int(rng.integers(0, 12))
        elif mode == 2:
            row = oracle.gen(2, 300 + i, i, 1).reshape(-1)[:sz].copy()
            row[sz // 2:] = row[:sz - sz // 2]
        elif mode == 3:
            row = oracle.gen(3, 300 + i, i, 1).reshape(-1)[:sz].copy()
            row[: sz // 3] = row[0]
        else:
            row = np.full(sz, int(rng.integers(0, 256)), np.uint8)
            for _ in range(int(rng.integers(0, 40))):
                row[int(rng.integers(0, sz))] = int(rng.integers(0, 256))
        blocks.append(row)
    res, dst = emu.encode(blocks, hc=True, groups=1, **{kind: True})
    for i, a in enumerate(blocks):
        want = oracle.compress(a, hc=True)
        assert res[i] == len(want), (i, a.size, res[i], len(want))
        assert np.array_equal(dst[i, :res[i]], want), (i, a.size)


def test_encode_hc_lane_slab_reuse(oracle):
    """The LZ4HC lane kernel re-zeroes its heads per block, sets chain[0] and leaves the rest of the previous block's chain
    (and, after a block > 64 KiB, its 32-bit heads where the 16-bit layout has its chain) in place, relying on every slot
    being written before a walk can reach it.  Here every lane encodes FIVE blocks in a row on one slab: lanes 0-3 a
    64 KiB block, a 70 000-byte block (32-bit heads), a 64 KiB block, a short one and another 64 KiB one, of mixed
    distributions; the other lanes five blocks of 300 - 9000 bytes, some with long repeats (the repeat optimisation
    rewires chains).  Every block must be the oracle's bytes."""
    rng = np.random.default_rng(17)
    rows = 5
    blocks = [None] * (64 * rows)
    big = [65536, 70000, 65536, 4000, 65536]
    for lane in range(64):
        for r in range(rows):
            if lane < 4:
                sz = big[r] - (lane * 7 if r != 1 else 0)
                dist = (2, 3, 2, 3)[(lane + r) % 4]
            else:
                sz = int(rng.integers(300, 9000))
                dist = 2 if (lane + r) % 3 else 3
            row = oracle.gen(dist, 100 + r, lane, (sz + 65535) // 65536).reshape(-1)[:sz].copy()
            if lane >= 4 and (lane + r) % 5 == 0:
                row[sz // 2:] = row[:sz - sz // 2]
            if lane >= 4 and (lane + r) % 7 == 0:
                row[: sz // 3] = 7                                    # a long run: every position hashes to one bucket
            blocks[r * 64 + lane] = row
    res, dst = emu.encode_hc_lane_static(blocks)
    for i, a in enumerate(blocks):
        want = oracle.compress(a, hc=True)
        assert res[i] == len(want), (i, a.size, res[i], len(want))
        assert np.array_equal(dst[i, :res[i]], want), (i, a.size)


def test_encode_hc_conv_slab_reuse(oracle):
    """The convergent LZ4HC kernel with ONE wavefront in its persistent grid: 64 lanes, 5 blocks each (handed out through the
    counter), 64 KiB / 70 000-byte (32-bit heads) / short blocks mixed, fuzzer-style and record-like, some with long repeats
    (the repeat optimisation) and long runs -- slabs are never re-initialised beyond the heads and chain[0]."""
    rng = np.random.default_rng(19)
    blocks = []
    for i in range(320):
        k = int(rng.integers(0, 40))
        sz = 65536 if k == 0 else 70000 if k == 1 else 65536 - int(rng.integers(1, 2000)) if k == 2 else int(rng.integers(300, 9000))
        row = oracle.gen(2 if i % 3 else 3, 200 + i % 7, i, (sz + 65535) // 65536).reshape(-1)[:sz].copy()
        if i % 6 == 0:
            row[sz // 2:] = row[:sz - sz // 2]
        if i % 11 == 0:
            row[: sz // 4] = 9
        blocks.append(row)
    res, dst = emu.encode(blocks, hc=True, conv=True, groups=1)
    for i, a in enumerate(blocks):
        want = oracle.compress(a, hc=True)
        assert res[i] == len(want), (i, a.size, res[i], len(want))
        assert np.array_equal(dst[i, :res[i]], want), (i, a.size)


def test_encode_lane_many_blocks_per_lane(oracle):
    """The lane encoder stamps its table entries with a per-lane block counter and zeroes the table only when the
    counter wraps (63 blocks) or after a block of the generic variant: a lane that encodes many blocks in a row,
    of both variants, must still produce the reference's bytes for every one of them."""
    rng = np.random.default_rng(11)
    sizes = [int(x) for x in rng.integers(13, 3000, 150)] + [65546, 70000, 300, 65547, 64, 5000]
    blocks = []
    for i, sz in enumerate(sizes):
        row = oracle.gen(2 if i % 3 else 3, 77, i, (sz + 65535) // 65536).reshape(-1)[:sz].copy()
        if i % 7 == 0:
            row[sz // 2:] = row[:sz - sz // 2]                    # long repeats: bucket reuse inside the block
        blocks.append(row)
    res, dst = emu.encode(blocks, lane=True, groups=1)             # 64 lanes, ~2.4 blocks per lane ...
    for i, a in enumerate(blocks):
        want = oracle.compress(a)
        assert res[i] == len(want) and np.array_equal(dst[i, :res[i]], want), (i, a.size)
    # ... and enough tiny blocks for every lane to wrap its 6-bit counter (64 lanes x > 63 blocks)
    tiny = []
    for i in range(64 * 70):
        sz = 13 + (i * 37) % 180
        row = oracle.gen(2, 5, i % 50, 1).reshape(-1)[(i % 97) * 7:(i % 97) * 7 + sz].copy()
        tiny.append(row)
    res, dst = emu.encode(tiny, lane=True, groups=1)
    for i in range(0, len(tiny), 13):
        want = oracle.compress(tiny[i])
        assert res[i] == len(want) and np.array_equal(dst[i, :res[i]], want), ("tiny", i, tiny[i].size)



@pytest.mark.parametrize("mapping", ["lane128s64"] + LANE3)
def test_lane_decoder_lockstep_lanes_and_copy_lengths(oracle, mapping):
    """64 identical blocks keep the 64 lanes of the lane-mapped decoder in lockstep, so every lane wants to flush in
    the same iteration (four rounds of the cooperative flush) -- on blocks built to contain matches of every length
    4..40 at offsets inside the ring, just behind it and far behind it (16- and 32-byte fetches), periodic matches and
    literal runs of 0..80 bytes."""
    rng = np.random.default_rng(23)
    data = bytearray(rng.integers(0, 256, 7000, dtype=np.uint8).tobytes())
    for off in (1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 60, 107, 108, 109, 110, 124, 125, 130, 200, 235, 236, 237, 238, 252, 253, 260, 500, 4097, 6000):
        for ml in list(range(4, 41)) + [64, 65, 100]:
            lit = int(rng.integers(0, 81)) if (off + ml) % 5 == 0 else int(rng.integers(0, 4))
            data += rng.integers(0, 256, lit, dtype=np.uint8).tobytes()
            start = len(data) - off
            for i in range(ml):                                   # byte-wise LZ77 copy (overlap allowed)
                data.append(data[start + i])
            data.append(int(rng.integers(0, 256)))                # break the match
    block = np.frombuffer(bytes(data), dtype=np.uint8)
    assert 30000 < block.size < 200000
    comp = oracle.compress(block)
    for known in (True, False):
        comps = [np.concatenate([comp, np.zeros(1024, np.uint8)]) for _ in range(64)]
        res, dst = emu.decode(comps, [block.size] * 64, known=known, src_lens=None if known else [len(comp)] * 64, **_mapping(mapping))
        for i in range(64):
            assert res[i] == (len(comp) if known else block.size), (known, i, res[i])
            assert np.array_equal(dst[i, :block.size], block), (known, i)


@pytest.mark.parametrize("gen", [3, 4])
def test_lane_decoder_starved_flush(oracle, gen):
    """The lane decoder with its cooperative flush cut down to 4 lines per round: with 64 busy lanes most of them miss
    flush rounds several times in a row, run their output rings full and sit out iterations -- also while the first or the
    next 16 bytes of a far match have already been fetched for them (those must be fetched again, not skipped).  Blocks:
    the lock-step block of the test above (far matches of every length) in 64 copies, and 64 different record-like /
    fuzzer-style blocks."""
    rng = np.random.default_rng(29)
    data = bytearray(rng.integers(0, 256, 5000, dtype=np.uint8).tobytes())
    for off in (130, 200, 236, 260, 500, 1000, 4097):
        for ml in list(range(4, 70, 3)):
            data += rng.integers(0, 256, int(rng.integers(0, 30)), dtype=np.uint8).tobytes()
            start = len(data) - off
            for i in range(ml):
                data.append(data[start + i])
            data.append(int(rng.integers(0, 256)))
    block = np.frombuffer(bytes(data), dtype=np.uint8)
    sets = [[block] * 64, [oracle.gen(3 if i % 2 else 2, 41, i, 1, 6000 + 37 * i)[0] for i in range(64)]]
    with emu.starved_flush():
        for blocks in sets:
            comps = [oracle.compress(a) for a in blocks]
            for known in (True, False):
                res, dst = emu.decode([np.concatenate([c, np.zeros(64, np.uint8)]) for c in comps], [a.size for a in blocks], known=known,
                                      src_lens=None if known else [len(c) for c in comps], lane=59192 if gen == 4 else 128, stage=64, gen=gen)
                for i, (a, c) in enumerate(zip(blocks, comps)):
                    assert res[i] == (len(c) if known else a.size), (known, i, res[i])
                    assert np.array_equal(dst[i, :a.size], a), (known, i)
        # the odd-but-legal and malformed streams and the error-code matrix, in the same starved state
        test_decode_arbitrary_streams(oracle, {2: "lane128s64", 3: "l3r128", 4: "l4c59192"}[gen])
        test_decode_error_codes_match_oracle(oracle, {2: "lane128s64", 3: "l3r128", 4: "l4c59192"}[gen])


@pytest.mark.parametrize("lane", [False] + LANE3, ids=["wave-per-block"] + LANE3)
def test_decode_arbitrary_streams(oracle, lane):
    """Streams that no encoder of ours produced (tests/stream_fuzz.py): whatever the oracle's decoders return for
    them -- bytes and return code, well formed or not -- the kernels return too, for both decoders, without touching
    a byte past the capacity."""
    import stream_fuzz
    cs = stream_fuzz.cases(2026, 240) + stream_fuzz.cases(2027, 24, max_size=12000)  # (the longer ones wrap the decoders' rings)
    comps = [c for (c, _), _ in cs]
    sizes = [t for _, t in cs]
    for (c, raw), t in cs:
        if raw is not None:
            assert raw.size == t
            ret, out = oracle.uncompress_raw(c, t)
            assert ret == len(c) and np.array_equal(out[:t], raw)          # the generator and the oracle agree on well-formed streams
    want_k = [oracle.uncompress_raw(c, t) for c, t in zip(comps, sizes)]
    pad = [np.concatenate([c, np.zeros(t + 1024, np.uint8)]) for c, t in zip(comps, sizes)]
    res, dst = emu.decode(pad, sizes, known=True, **_mapping(lane))
    for i, (w, out) in enumerate(want_k):
        assert res[i] == w, ("known", i, res[i], w)
        if w >= 0:
            assert np.array_equal(dst[i, :sizes[i]], out[:sizes[i]]), ("known", i)
        assert (dst[i, sizes[i]:] == 0xA5).all(), ("known canary", i)
    caps = [t + (i % 3) * 7 - (5 if i % 11 == 0 else 0) for i, t in enumerate(sizes)]
    want_u = [oracle.uncompress_unknown_raw(c, len(c), cap) for c, cap in zip(comps, caps)]
    padu = [np.concatenate([c, np.zeros(8, np.uint8)]) for c in comps]
    res, dst = emu.decode(padu, caps, known=False, src_lens=[len(c) for c in comps], **_mapping(lane))
    for i, (w, out) in enumerate(want_u):
        assert res[i] == w, ("unknown", i, res[i], w)
        if w >= 0:
            assert np.array_equal(dst[i, :w], out[:w]), ("unknown", i)
        assert (dst[i, max(caps[i], 0):] == 0xA5).all(), ("unknown canary", i)
