"""-m gpu parity tests proper: the HIP path, called through the C ABI, against the CPU oracle and the
committed golden vectors (generated from the reference's own C).  Bit-exact for every byte and every
return code.  Mirrors the reference's test strategy: cross-implementation identity
(ConformanceTests.cs:59-68,121-148), round trips, and upstream's fuzzer matrix (original/fuzzer.c:176-227)."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vectors.json")))
SIZES = (0, 1, 12, 13, 14, 64, 65, 300, 4096, 20000, 65535, 65536)


def sha(a):
    return hashlib.sha256(bytes(a)).hexdigest()


@pytest.fixture(scope="module")
def gpu():
    import gpu_helpers
    from lz4net_amd import _lib
    assert _lib.lib().lz4hip_device_count() >= 1, "no HIP device visible"
    name = _lib.lib().lz4hip_codec_name().decode()
    assert "gfx950" in name, name
    return gpu_helpers


def _blocks(oracle, sizes=SIZES, seeds=(5,)):
    out = []
    for dist in range(4):
        for seed in seeds:
            for n in sizes:
                out.append(oracle.gen(dist, seed, n, 1, max(n, 1))[0][:n])
    rng = np.random.default_rng(7)
    for n in (50, 700, 9000, 65536):
        for k in (2, 3, 16):
            out.append(rng.integers(0, k, n, dtype=np.uint8))
    out.append(np.frombuffer(b"abcabcabcabcabcabcabcabcabcabc" * 40, dtype=np.uint8))
    return out


def test_golden_synth_vectors(gpu, oracle):
    # known answers from the reference's own C (tests/golden/make_golden.py); no oracle in the loop
    es = GOLD["synth"]
    blocks = [oracle.gen(e["dist"], e["seed"], e["block"], 1, max(e["n"], 1))[0][:e["n"]] for e in es]
    for hc, klen, ksha in ((False, "fast_len", "fast_sha256"), (True, "hc_len", "hc_sha256")):
        res, dst = gpu.encode(blocks, hc=hc)
        for i, e in enumerate(es):
            assert res[i] == e[klen], (i, hc, e["dist"], e["n"], res[i], e[klen])
            assert sha(dst[i, :res[i]]) == e[ksha], (i, hc, e["dist"], e["n"])


from conftest import ForcedMapping, _forced_fixture  # noqa: E402


@pytest.fixture(params=["wave", "lane"])
def encoder(request):
    """Both block->hardware mappings of the fast encoder (lz4hip_encode.hpp / lz4hip_encode_lane.hpp); the fixture
    asserts through lz4hip_dispatch_counts that the mapping named is the one that ran."""
    yield from _forced_fixture("LZ4HIP_ENCODER", request.param)


def test_fast_encode_bit_exact(gpu, oracle, encoder):
    blocks = _blocks(oracle, sizes=SIZES + (65546, 65547, 70000, 200000), seeds=(5, 6))
    res, dst = gpu.encode(blocks)
    for i, a in enumerate(blocks):
        want = oracle.compress(a)
        assert res[i] == len(want), (i, a.size, res[i], len(want))
        assert np.array_equal(dst[i, :res[i]], want), (i, a.size)
        assert (dst[i, a.size + a.size // 255 + 16:] == 0xA5).all()


def test_fast_encode_wave_second_version_paths(gpu, oracle):
    """The second version of the wavefront-mapped 64k encoder (lz4hip_encode.hpp, encode_fast_block64k; round 6) on the GPU: its register
    window (re-based every ~244 bytes and after long matches), the combined candidate test + 256-byte count, its continuation past 256
    equal bytes and the byte-wise form near the end of a block, the merged catch-up + count loads, the one-store sequence emit next to the
    general one -- sizes around every threshold, dense / sparse / periodic / long-run data, exact and too-small output limits.  The twin of
    tests/test_simt_emulation.py::test_encode_fast64k_window_wide_count_and_single_store_paths, through the C ABI with the mapping forced."""
    from lz4net_amd import _lib
    rng = np.random.default_rng(606)
    blocks = []
    for n in list(range(13, 40)) + [255, 256, 257, 260, 261, 262, 263, 268, 269, 270, 300, 511, 512, 513, 517, 518, 519, 777, 1200, 5000, 33333, 65536, 65546]:
        for dist in (2, 3):
            blocks.append(oracle.gen(dist, 600 + n, n, 1, n)[0][:n])
    for n in (300, 1000, 4000, 20000, 65536):
        blocks.append(rng.integers(0, 2, n, dtype=np.uint8))
        blocks.append(np.tile(rng.integers(0, 256, 7, dtype=np.uint8), n // 7 + 1)[:n])
        a = np.tile(rng.integers(0, 256, 300, dtype=np.uint8), n // 300 + 1)[:n].copy()
        a[rng.integers(0, n, max(n // 900, 1))] ^= 0x55
        blocks.append(a)
        b = rng.integers(0, 256, n, dtype=np.uint8)
        for _ in range(n // 200):
            src, ln, dstp = int(rng.integers(0, n - 40)), int(rng.integers(4, 40)), int(rng.integers(40, n - 40))
            b[dstp:dstp + ln] = b[src:src + ln]
        blocks.append(b)
    want = [oracle.compress(a) for a in blocks]
    with ForcedMapping("LZ4HIP_ENCODER", "wave"):
        for wg5 in (1, 2, 0):           # one block per workgroup, five per workgroup (the form of batches > nine blocks per CU; here with a ragged last workgroup), the default
            with _lib.tuning(encoder_wg5=wg5):
                res, dst = gpu.encode(blocks if wg5 != 2 else blocks[:len(blocks) - (len(blocks) % 5 == 0)])
            for i, (a, w) in enumerate(zip(blocks[:len(res)], want)):
                assert res[i] == len(w), (wg5, i, a.size, res[i], len(w))
                assert np.array_equal(dst[i, :res[i]], w), (wg5, i, a.size)
                assert (dst[i, a.size + a.size // 255 + 16:] == 0xA5).all()
        for delta in (0, -1, -3, -9):
            caps = [max(len(w) + delta, 0) for w in want]
            res, dst = gpu.encode(blocks, caps=caps)
            for i, a in enumerate(blocks):
                r = oracle.compress_raw(a, caps[i])[0]
                assert res[i] == r, (i, a.size, delta, res[i], r)
                assert (dst[i, caps[i]:] == 0xA5).all(), (i, delta, "wrote past the capacity")
                if r > 0:
                    assert np.array_equal(dst[i, :r], want[i]), (i, delta)


@pytest.fixture(params=["wave", "lane"])
def hc_mapping(request):
    yield from _forced_fixture("LZ4HIP_HC", request.param)


def test_hc_encode_bit_exact(gpu, oracle, hc_mapping):
    blocks = _blocks(oracle, sizes=SIZES + (65537, 70000, 150000), seeds=(5,))
    res, dst = gpu.encode(blocks, hc=True)
    for i, a in enumerate(blocks):
        want = oracle.compress(a, hc=True)
        assert res[i] == len(want), (i, a.size, res[i], len(want))
        assert np.array_equal(dst[i, :res[i]], want), (i, a.size)


def test_limited_output(gpu, oracle, encoder, hc_mapping):
    # original/fuzzer.c:212-227: exact capacity succeeds, one byte less returns 0, canary untouched
    blocks = _blocks(oracle, sizes=(13, 300, 4096, 65536))
    for hc in (False, True):
        lens = [len(oracle.compress(a, hc=hc)) for a in blocks]
        for delta in (0, -1, -7):
            caps = [max(l + delta, 0) for l in lens]
            res, dst = gpu.encode(blocks, caps=caps, hc=hc)
            for i, a in enumerate(blocks):
                want = oracle.compress_raw(a, caps[i], hc=hc)[0]
                assert res[i] == want, (i, hc, delta, res[i], want)
                assert (dst[i, caps[i]:] == 0xA5).all(), (i, hc, delta, "wrote past the capacity")


@pytest.fixture(params=["wave", "lane"])
def decoder(request):
    """Both block->hardware mappings of the decoder (lz4hip_decode.hpp / lz4hip_decode_lane4.hpp).  The override
    holds for every batch size, and the fixture asserts through lz4hip_dispatch_counts that the mapping named ran
    and the other one did not."""
    yield from _forced_fixture("LZ4HIP_DECODER", request.param)


def test_decode_known_and_unknown(gpu, oracle, decoder):
    blocks = _blocks(oracle)
    for hc in (False, True):
        comps = [oracle.compress(a, hc=hc) for a in blocks]
        res, dst = gpu.decode(comps, [a.size for a in blocks], known=True)
        for i, (a, c) in enumerate(zip(blocks, comps)):
            assert res[i] == len(c), (i, hc, res[i], len(c))
            assert np.array_equal(dst[i, :a.size], a), (i, hc)
            assert (dst[i, a.size:] == 0xA5).all()
        for extra in (0, 1, 100):
            res, dst = gpu.decode(comps, [a.size + extra for a in blocks], known=False)
            for i, a in enumerate(blocks):
                assert res[i] == a.size, (i, hc, extra, res[i])
                assert np.array_equal(dst[i, :a.size], a)
                assert (dst[i, a.size + extra:] == 0xA5).all()


def test_decode_error_codes(gpu, oracle, decoder):
    rng = np.random.default_rng(11)
    blocks = _blocks(oracle, sizes=(13, 300, 4096, 65536))
    comps = [oracle.compress(a) for a in blocks]
    cases_k, want_k, cases_u, want_u = [], [], [], []
    for a, c in zip(blocks, comps):
        for osize in (a.size - 1, a.size + 1, a.size // 2):
            cases_k.append((c, osize)); want_k.append(oracle.uncompress_raw(c, osize)[0])
        for isz, mo in ((len(c), a.size - 1), (len(c) - 1, a.size), (len(c) + 1, a.size), (0, a.size)):
            cases_u.append((c, isz, mo)); want_u.append(oracle.uncompress_unknown_raw(c, isz, mo)[0])
        for _ in range(6):
            cc = c.copy()
            cc[rng.integers(0, len(c))] = rng.integers(0, 256)
            cases_k.append((cc, a.size)); want_k.append(oracle.uncompress_raw(cc, a.size)[0])
            cases_u.append((cc, len(cc), a.size)); want_u.append(oracle.uncompress_unknown_raw(cc, len(cc), a.size)[0])
    pad = [np.concatenate([c, np.zeros(max(o, 0) + 1024, np.uint8)]) for c, o in cases_k]
    res, dst = gpu.decode(pad, [o for _, o in cases_k], known=True)
    for i, w in enumerate(want_k):
        assert res[i] == w, ("known", i, res[i], w)
        assert (dst[i, max(cases_k[i][1], 0):] == 0xA5).all()
    padu = [np.concatenate([c, np.zeros(8, np.uint8)]) for c, _, _ in cases_u]
    res, dst = gpu.decode(padu, [mo for _, _, mo in cases_u], known=False, src_lens=[i for _, i, _ in cases_u])
    for i, w in enumerate(want_u):
        assert res[i] == w, ("unknown", i, res[i], w)
        assert (dst[i, max(cases_u[i][2], 0):] == 0xA5).all()


def test_fuzzer_matrix(gpu, oracle, decoder, encoder):
    # original/fuzzer.c:149-227 on 32 KiB fuzzer-generated buffers, batched
    LEN, N = 1 << 15, 96
    blocks = [oracle.gen(2, 99, i, 1, LEN)[0] for i in range(N)]
    bound = LEN + LEN // 255 + 16
    rh, dh = gpu.encode(blocks, hc=True)
    rf, df = gpu.encode(blocks, hc=False)
    assert (rh > 0).all() and (rf > 0).all()
    comps = [df[i, :rf[i]].copy() for i in range(N)]
    for i in range(N):
        assert np.array_equal(comps[i], oracle.compress(blocks[i])), i
        assert np.array_equal(dh[i, :rh[i]], oracle.compress(blocks[i], hc=True)), i
    r, d = gpu.decode(comps, [LEN] * N, known=True)
    assert (r == rf).all() and all(np.array_equal(d[i, :LEN], blocks[i]) for i in range(N))
    pad = [np.concatenate([c, np.zeros(LEN + 1024, np.uint8)]) for c in comps]
    assert (gpu.decode(pad, [LEN - 1] * N, known=True)[0] < 0).all()      # one byte missing => must fail
    assert (gpu.decode(pad, [LEN + 1] * N, known=True)[0] < 0).all()      # one byte too much => must fail
    assert (gpu.decode(comps, [LEN + 1] * N, known=False)[0] == LEN).all()
    assert (gpu.decode(comps, [LEN] * N, known=False)[0] == LEN).all()
    assert (gpu.decode(comps, [LEN - 1] * N, known=False)[0] < 0).all()
    padu = [np.concatenate([c, np.zeros(8, np.uint8)]) for c in comps]
    assert (gpu.decode(padu, [LEN] * N, known=False, src_lens=[len(c) - 1 for c in comps])[0] < 0).all()
    assert (gpu.decode(padu, [LEN] * N, known=False, src_lens=[len(c) + 1 for c in comps])[0] < 0).all()
    # compress into exactly the needed size works, one byte less returns 0 without touching the canary
    r, d = gpu.encode(blocks, caps=list(rf))
    assert (r == rf).all() and all((d[i, rf[i]:] == 0xA5).all() for i in range(N))
    r, d = gpu.encode(blocks, caps=[x - 1 for x in rf])
    assert (r == 0).all() and all((d[i, rf[i] - 1:] == 0xA5).all() for i in range(N))
    r, d = gpu.encode(blocks, caps=list(rh), hc=True)
    assert (r == rh).all()
    r, d = gpu.encode(blocks, caps=[x - 1 for x in rh], hc=True)
    assert (r == 0).all() and all((d[i, rh[i] - 1:] == 0xA5).all() for i in range(N))
    # issue 52 overflow regression (original/fuzzer.c:96-117): a 16 MiB run of 0xFF length bytes
    bad = np.full(16840000, 0xFF, np.uint8)
    bad[:3] = (0x0F, 0, 0)
    assert gpu.decode([bad], [20 << 20], known=True)[0][0] < 0


def test_lz4codec_api(gpu, oracle):
    # the reference's load-time self test (src/LZ4/LZ4Codec.cs:173-239): Lorem x5, fast + HC,
    # known + unknown length; plus WrapTests (src/LZ4.Tests/WrapTests.cs:11-48)
    from lz4net_amd import LZ4Codec
    from lz4net_amd.codec import ArgumentException, ArgumentNullException
    e = next(x for x in GOLD["inline"] if x["name"] == "lorem_x5")
    original = bytes.fromhex(e["input_hex"])
    for enc, key in ((LZ4Codec.Encode, "fast_hex"), (LZ4Codec.EncodeHC, "hc_hex")):
        comp = enc(original, 0, len(original))
        assert comp.hex() == e[key]
        out = bytearray(len(original))
        assert LZ4Codec.Decode(comp, 0, len(comp), out, 0, len(out), True) == len(original)
        assert bytes(out) == original
        out = bytearray(len(original) + 50)
        assert LZ4Codec.Decode(comp, 0, len(comp), out, 0, len(out), False) == len(original)
        assert bytes(out[:len(original)]) == original
        assert LZ4Codec.Decode(comp, 0, len(comp), len(original)) == original
        with pytest.raises(ArgumentException):
            LZ4Codec.Decode(comp, 0, len(comp) - 1, bytearray(len(original)), 0, len(original), True)
    assert LZ4Codec.MaximumOutputLength(65536) == 65809
    assert "gfx950" in LZ4Codec.CodecName
    assert LZ4Codec.Encode(b"", 0, 0, bytearray(10), 0, 10) == 0              # inputLength == 0 => 0
    with pytest.raises(ArgumentNullException):
        LZ4Codec.Encode(None, 0, 5, bytearray(10), 0, 10)
    with pytest.raises(ArgumentException):
        LZ4Codec.Encode(b"abc", 2, 5, bytearray(10), 0, 10)
    assert LZ4Codec.Encode(original, 0, len(original), bytearray(10), 0, 10) == 0   # too small => 0
    assert LZ4Codec.EncodeHC(original, 0, len(original), bytearray(10), 0, 10) == -1  # HC maps <=0 to -1
    lorem4 = bytes.fromhex(next(x for x in GOLD["inline"] if x["name"] == "lorem_x1")["input_hex"]) * 4
    rnd = bytes(np.random.default_rng(0).integers(0, 256, 2048, dtype=np.uint8))
    for wrap in (LZ4Codec.Wrap, LZ4Codec.WrapHC):
        w = wrap(lorem4)
        assert len(w) < len(lorem4) and LZ4Codec.Unwrap(w) == lorem4
        w = wrap(rnd)
        assert len(w) == len(rnd) + 8 and LZ4Codec.Unwrap(w) == rnd       # incompressible: stored raw
        assert LZ4Codec.Unwrap(wrap(b"x")) == b"x"
        assert wrap(b"") == bytes(8)


def test_lz4h_shaped_single_block_calls(gpu, oracle):
    import ctypes as C
    from lz4net_amd import _lib
    L = _lib.lib()
    a = oracle.gen(3, 4, 0, 1, 65536)[0]
    out = np.zeros(65809 + 1, np.uint8)
    n = L.lz4hip_compress(a.ctypes.data, out.ctypes.data, a.size)
    want = oracle.compress(a)
    assert n == len(want) and np.array_equal(out[:n], want)
    nh = L.lz4hip_compressHC(a.ctypes.data, out.ctypes.data, a.size)
    wanth = oracle.compress(a, hc=True)
    assert nh == len(wanth) and np.array_equal(out[:nh], wanth)
    back = np.zeros(a.size, np.uint8)
    comp = np.ascontiguousarray(want)
    assert L.lz4hip_uncompress(comp.ctypes.data, back.ctypes.data, a.size) == len(want)   # no source length given
    assert np.array_equal(back, a)
    back[:] = 0
    assert L.lz4hip_uncompress_unknownOutputSize(comp.ctypes.data, back.ctypes.data, len(want), a.size) == a.size
    assert np.array_equal(back, a)


def test_host_batch_many_slices_and_layouts(gpu, oracle):
    """The host-pointer entry points cut a batch into ~64 MiB slices staged through pinned memory (gather/scatter on
    host threads): a batch spanning several slices, with ragged and empty rows and an offset-addressed (packed) layout,
    must give exactly what the oracle gives block by block."""
    import ctypes as C
    from lz4net_amd import _lib
    rng = np.random.default_rng(17)
    n = 2600                                                     # ~170 MiB of input -> several slices each way
    sizes = rng.integers(0, 65537, n)
    sizes[:8] = [0, 1, 12, 13, 65536, 65535, 64, 4096]
    raw_rows = oracle.gen(2, 99, 0, n)
    # packed source: block i at src_off[i], lengths ragged
    src_off = np.concatenate(([0], np.cumsum(sizes[:-1]))).astype(np.int64)
    src = np.concatenate([raw_rows[i, :sizes[i]] for i in range(n)] + [np.zeros(16, np.uint8)])
    lens = sizes.astype(np.int32)
    caps = np.array([int(s) + int(s) // 255 + 16 for s in sizes], np.int32)
    dst_off = np.concatenate(([0], np.cumsum(caps[:-1].astype(np.int64)))).astype(np.int64)
    dst = np.full(int(caps.astype(np.int64).sum()) + 64, 0xA5, np.uint8)
    res = np.zeros(n, np.int32)
    b = _lib.Batch(src=src.ctypes.data, src_off=src_off.ctypes.data, src_stride=0, src_len=lens.ctypes.data,
                   dst=dst.ctypes.data, dst_off=dst_off.ctypes.data, dst_stride=0, dst_cap=caps.ctypes.data,
                   dst_cap_all=0, src_len_all=0, result=res.ctypes.data, n_blocks=n)
    _lib.tuning_set("host_slices", 5)                            # (a batch this small would go in one piece by default)
    _lib.check(_lib.lib().lz4hip_encode_batch_host(C.byref(b), 0))
    for i in list(range(16)) + list(range(16, n, 97)) + [n - 1]:
        ret, out = oracle.compress_raw(raw_rows[i, :sizes[i]], int(caps[i]))
        assert res[i] == ret, i
        assert bytes(dst[dst_off[i]:dst_off[i] + ret]) == bytes(out[:ret]), i
    assert (dst[-64:] == 0xA5).all()
    # decode the packed compressed rows back into a strided destination, known sizes
    back = np.full((n, 65536 + 32), 0x5A, np.uint8)
    used = np.zeros(n, np.int32)
    d = _lib.Batch(src=dst.ctypes.data, src_off=dst_off.ctypes.data, src_stride=0, src_len=res.ctypes.data,
                   dst=back.ctypes.data, dst_off=None, dst_stride=back.strides[0], dst_cap=lens.ctypes.data,
                   dst_cap_all=0, src_len_all=0, result=used.ctypes.data, n_blocks=n)
    _lib.check(_lib.lib().lz4hip_decode_batch_host(C.byref(d), 1))
    _lib.tuning_set("host_slices", 0)
    assert (used == res).all()
    for i in range(n):
        assert np.array_equal(back[i, :sizes[i]], raw_rows[i, :sizes[i]]), i
        assert (back[i, sizes[i]:] == 0x5A).all(), i
    # ... and with the default cut (one piece for a batch of this size)
    used[:] = 0; back[:] = 0x5A
    _lib.check(_lib.lib().lz4hip_decode_batch_host(C.byref(d), 1))
    assert (used == res).all() and all(np.array_equal(back[i, :sizes[i]], raw_rows[i, :sizes[i]]) for i in range(0, n, 13))


def test_host_fast_encode_default_slices_of_one_residency_round(gpu, oracle):
    """A host-pointer fast encode of 4 096 blocks or more is cut, by default, into the fewest EQUAL slices of at most one residency round of the
    wavefront-mapped encoder (ten blocks per CU; lz4hip_api.hip: encode_host_slice_blocks) and runs as one pipeline; a slice's copies move only the
    rows it has.  5 203 blocks (three slices, the last one shorter) with some ragged rows: every result and a sample of the payloads against the
    oracle, canaries behind every row, and the same call with two pipelines (knob host_workers) gives the same bytes."""
    import ctypes as C
    from lz4net_amd import _lib
    n = 5203
    raw = oracle.gen(2, 4242, 0, n)
    lens = np.full(n, 65536, np.int32)
    lens[[0, 1, 2, 1733, 1734, 1735, 3467, 3468, n - 1]] = [0, 13, 65535, 12, 300, 65536, 4096, 1, 777]
    cap = 65536 + 65536 // 255 + 16
    caps = np.full(n, cap, np.int32)
    outs = []
    for workers in (0, 2):
        dst = np.full((n, cap + 48), 0xA5, np.uint8)
        res = np.full(n, -7, np.int32)
        b = _lib.Batch(src=raw.ctypes.data, src_off=None, src_stride=raw.strides[0], src_len=lens.ctypes.data, dst=dst.ctypes.data, dst_off=None,
                       dst_stride=dst.strides[0], dst_cap=caps.ctypes.data, dst_cap_all=0, src_len_all=65536, result=res.ctypes.data, n_blocks=n)
        with _lib.tuning(host_workers=workers):
            _lib.check(_lib.lib().lz4hip_encode_batch_host(C.byref(b), 0))
        assert (dst[:, cap:] == 0xA5).all(), workers
        outs.append((res, dst))
    res, dst = outs[0]
    assert np.array_equal(res, outs[1][0])
    for i in range(n):
        assert np.array_equal(dst[i, :res[i]], outs[1][1][i, :res[i]]), i
    for i in [0, 1, 2, 3, 1732, 1733, 1734, 1735, 1736, 3466, 3467, 3468, 3469, n - 2, n - 1] + list(range(7, n, 211)):
        ret, out = oracle.compress_raw(raw[i, :lens[i]], cap)
        assert res[i] == ret, (i, res[i], ret)
        assert bytes(dst[i, :ret]) == bytes(out[:ret]), i


def test_host_entry_points_from_several_threads(gpu, oracle):
    """The host-pointer entry points keep their staging (pinned buffers, copy streams, events) per calling thread and
    share only the leased device workspaces: four threads encoding and decoding their own batches at the same time
    must each get the oracle's bytes."""
    import threading
    import gpu_helpers as gh
    errors = []

    def worker(t):
        try:
            rng = np.random.default_rng(100 + t)
            for rep in range(3):
                blocks = [oracle.gen(2 + (t + i) % 2, 300 + t, 7 * rep + i, 1)[0][:int(rng.integers(1, 65537))] for i in range(40)]
                hc = (t + rep) % 2 == 1
                res, dst = gh.encode(blocks, hc=hc)
                comps = []
                for i, a in enumerate(blocks):
                    want = oracle.compress(a, hc=hc)
                    if res[i] != len(want) or not np.array_equal(dst[i, :res[i]], want):
                        errors.append(("enc", t, rep, i))
                    comps.append(want)
                used, back = gh.decode(comps, [a.size for a in blocks], known=True)
                for i, a in enumerate(blocks):
                    if used[i] != len(comps[i]) or not np.array_equal(back[i, :a.size], a):
                        errors.append(("dec", t, rep, i))
        except Exception as e:
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]


def test_lane_decoder_lockstep_lanes_and_copy_lengths(gpu, oracle, monkeypatch):
    """GPU twin of the emulator test of the same name: 256 identical crafted blocks through the lane mapping (every
    lane flushes in the same iteration; matches of every length 4..40 at offsets inside / just behind / far behind
    the LDS ring, periodic matches, literal runs of 0..80 bytes), known and unknown output size."""
    rng = np.random.default_rng(23)
    data = bytearray(rng.integers(0, 256, 7000, dtype=np.uint8).tobytes())
    for off in (1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 60, 107, 108, 109, 110, 124, 125, 130, 200, 235, 236, 237, 238, 252, 253, 260, 500, 4097, 6000):
        for ml in list(range(4, 41)) + [64, 65, 100]:
            lit = int(rng.integers(0, 81)) if (off + ml) % 5 == 0 else int(rng.integers(0, 4))
            data += rng.integers(0, 256, lit, dtype=np.uint8).tobytes()
            start = len(data) - off
            for i in range(ml):
                data.append(data[start + i])
            data.append(int(rng.integers(0, 256)))
    block = np.frombuffer(bytes(data), dtype=np.uint8)
    comp = oracle.compress(block)
    n = 256
    with ForcedMapping("LZ4HIP_DECODER", "lane"):
        used, back = gpu.decode([comp] * n, [block.size] * n, known=True)
        assert (used == len(comp)).all() and all(np.array_equal(back[i, :block.size], block) for i in range(n))
        produced, back = gpu.decode([comp] * n, [block.size + 9] * n, known=False)
        assert (produced == block.size).all() and all(np.array_equal(back[i, :block.size], block) for i in range(n))


def test_decode_arbitrary_streams(gpu, oracle, decoder):
    """GPU twin of the emulator test: streams no encoder of ours produced (tests/stream_fuzz.py), well formed or
    not -- same bytes and same return codes as the oracle for both decoders, every mapping, nothing written past the
    capacity."""
    import stream_fuzz
    cs = stream_fuzz.cases(99, 600) + stream_fuzz.cases(100, 150, max_size=20000)    # (the longer ones wrap the decoders' rings)
    comps = [c for (c, _), _ in cs]
    sizes = [t for _, t in cs]
    pad = [np.concatenate([c, np.zeros(t + 1024, np.uint8)]) for c, t in zip(comps, sizes)]
    res, dst = gpu.decode(pad, sizes, known=True)
    # (a match with offset 0 keeps what the destination held: through the host-pointer entry points that is the device staging
    #  image, not the caller's buffer, so the BYTES of such streams are not comparable here -- return codes and canaries are;
    #  the emulator twin compares the bytes as well)
    holes = [stream_fuzz.has_zero_offset(c, t + 8) for c, t in zip(comps, sizes)]
    for i, (c, t) in enumerate(zip(comps, sizes)):
        w, out = oracle.uncompress_raw(c, t)
        assert res[i] == w, ("known", i, res[i], w)
        if w >= 0 and not holes[i]:
            assert np.array_equal(dst[i, :t], out[:t]), ("known", i)
        assert (dst[i, t:] == 0xA5).all(), ("known canary", i)
    caps = [t + (i % 3) * 7 - (5 if i % 11 == 0 else 0) for i, t in enumerate(sizes)]
    padu = [np.concatenate([c, np.zeros(8, np.uint8)]) for c in comps]
    res, dst = gpu.decode(padu, caps, known=False, src_lens=[len(c) for c in comps])
    for i, (c, cap) in enumerate(zip(comps, caps)):
        w, out = oracle.uncompress_unknown_raw(c, len(c), cap)
        assert res[i] == w, ("unknown", i, res[i], w)
        if w >= 0 and not holes[i]:
            assert np.array_equal(dst[i, :w], out[:w]), ("unknown", i)
        assert (dst[i, max(cap, 0):] == 0xA5).all(), ("unknown canary", i)


def test_lds_out_of_range_stores_are_dropped(gpu, tmp_path):
    """The hardware rule the lane decoder's appends rest on since round 5 (lz4hip_decode_lane4.hpp, LZ4HIP_DEC4_DUAL_STORE): a DS store
    whose address lies outside the workgroup's LDS allocation is DROPPED -- no fault, no word of any co-resident workgroup's allocation
    changes.  tools/lds_out_of_range.hip checks exactly that (twelve workgroups of the decoder's 12 800 bytes resident per CU, 2 000 rounds
    of out-of-range stores per lane, past the end and below address 0); a device on which it fails would corrupt decoded bytes, and
    this test says why before the decoder tests do."""
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # (a GPU box without hipcc cannot run the stand-alone check: that is a FAILURE of this test, not a skip -- the library's own load-time
    #  probe, test_decoder_dual_store_is_a_checked_precondition below, is then the only evidence and this line says so)
    assert os.path.exists(hipcc), "no hipcc on this GPU box: tools/lds_out_of_range.hip cannot be built here"
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "lds_out_of_range.hip")
    exe = str(tmp_path / "lds_out_of_range")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", src, "-o", exe], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "pattern words changed in any workgroup's allocation: 0 " in r.stdout


def test_decoder_fuzz_slice(gpu, oracle):
    """A 30-second slice of tools/fuzz_gpu_decoders.py in every -m gpu run (tests/decoder_fuzz.py): arbitrary streams x known / unknown
    output size x the four decoder forms -- wavefront mapping with bursts, lane mapping in workgroups of one and of four wavefronts, the persistent lane grid with ONE wavefront
    (every lane restarts with its predecessor's loads still in flight) -- against the CPU oracle: results, bytes, canaries.
    ALWAYS the same three seeds (1000..1002 with 240 streams each: a failure here reproduces on any box with
    `decoder_fuzz.run_seed(oracle, seed, 240)`), then, as an extra while the 30 seconds last, seeds that move with the day so that
    successive runs also cover new streams; every message names its seed and `per`."""
    import time
    import decoder_fuzz
    per = 240
    msgs = []
    total, bad, done = decoder_fuzz.run(oracle, 1000, 3, per, seconds=None, report=msgs.append)
    print(f"decoder fuzz slice: fixed seeds 1000..1002, per={per}: {total} comparisons, {bad} mismatches")
    assert done == 3 and bad == 0, (f"fixed seeds 1000..1002, per={per}; replay: tests/decoder_fuzz.py run_seed(oracle, seed, {per})", msgs[:10])
    assert total >= 3 * 270 * 2 * 4, (total, f"fixed seeds 1000..1002, per={per}")
    first = 2000 + int(time.time() // 86400) % 1000 * 16
    t0 = time.time()
    extra_total = extra_bad = extra_done = 0
    for seed in range(first, first + 16):
        if time.time() - t0 > 20.0:
            break
        t, b = decoder_fuzz.run_seed(oracle, seed, per, msgs.append)
        extra_total += t; extra_bad += b; extra_done += 1
    print(f"decoder fuzz slice: day-rotating seeds {first}..{first + extra_done - 1}, per={per}: {extra_total} comparisons, {extra_bad} mismatches")
    assert extra_bad == 0, (f"day-rotating seeds {first}..{first + extra_done - 1}, per={per}; replay: python tools/fuzz_gpu_decoders.py (seed range above)", msgs[:10])


def test_decoder_dual_store_is_a_checked_precondition(gpu, oracle):
    """The lane decoder's default instantiation stores every ring row twice and relies on gfx950 dropping the LDS store that falls outside the
    workgroup's allocation (lz4hip_decode_lane4.hpp, L4_APPEND).  The library checks that rule ITSELF, once per device, before the first
    lane-mapped decode (lds_drop_probe_kernel) and falls back to the instantiation that wraps its rows where the probe does not confirm it.
    Here: the probe's verdict on this device is 'confirmed' (read-only knob decoder_dual_store), and the fallback -- forced through the knob
    decoder_wrapped_stores -- produces the same results and bytes as the default on well-formed, truncated and corrupt streams, in both
    forms of the lane kernel (one block per lane; the persistent grid with one wavefront)."""
    from lz4net_amd import _lib
    assert _lib.tuning_get("decoder_dual_store") == 1, "this gfx950 device did not confirm that out-of-range LDS stores are dropped"
    rng = np.random.default_rng(20260930)
    blocks = [oracle.gen(d, 31, i, 1, n)[0] for d in (2, 3) for i, n in enumerate((65536, 65536, 40000, 4097, 300, 65536))]
    comps = [oracle.compress(a) for a in blocks]
    sizes = [a.size for a in blocks]
    # damaged copies: truncated, and with a few bytes overwritten
    for k in range(8):
        c = comps[k % len(blocks)].copy()
        if k % 2:
            c = c[: max(1, c.size - 1 - int(rng.integers(0, 200)))]
        else:
            for _ in range(3):
                c[int(rng.integers(0, c.size))] = int(rng.integers(0, 256))
        comps.append(c); sizes.append(blocks[k % len(blocks)].size)
    outs = {}
    for wrapped in (0, 1):
        for persist, groups in ((2, 0), (1, 1)):
            with _lib.tuning(decoder="lane", decoder_wrapped_stores=wrapped, decoder_persist=persist, decoder_groups=groups):
                assert _lib.tuning_get("decoder_dual_store") == (0 if wrapped else 1)
                before = _lib.dispatch_counts()[_lib.K_DECODE_LANE]
                used, back = gpu.decode(comps, sizes, known=True)
                produced, back_u = gpu.decode(comps, [s + 5 for s in sizes], known=False)
                assert _lib.dispatch_counts()[_lib.K_DECODE_LANE] >= before + 2
            outs[(wrapped, persist)] = (np.array(used), back.copy(), np.array(produced), back_u.copy())
    for i, c in enumerate(comps):                                      # every variant = the default form, damaged streams included
        wu, _ = oracle.uncompress_unknown_raw(c, len(c), sizes[i] + 5)
        for key, (used, back, produced, back_u) in outs.items():
            ref = outs[(0, 2)]
            assert produced[i] == wu, (key, i, produced[i], wu)
            assert used[i] == ref[0][i] and produced[i] == ref[2][i], (key, i)
            if used[i] >= 0:
                assert np.array_equal(back[i, :sizes[i]], ref[1][i, :sizes[i]]), (key, i)
            if produced[i] >= 0:
                assert np.array_equal(back_u[i, :produced[i]], ref[3][i, :produced[i]]), (key, i)
    for i, a in enumerate(blocks):                                     # the undamaged streams decode to their input
        assert outs[(1, 2)][0][i] == len(comps[i]) and np.array_equal(outs[(1, 2)][1][i, :a.size], a), i
        assert outs[(1, 1)][2][i] == a.size and np.array_equal(outs[(1, 1)][3][i, :a.size], a), i


def test_release_workspaces_then_reuse(gpu, oracle):
    """lz4hip_release_workspaces() gives back the cached kernel workspaces and the calling thread's host-pointer staging;
    the next calls simply allocate again and produce the same bytes."""
    from lz4net_amd import _lib
    blocks = [oracle.gen(2, 77, i, 1, 4096 + 11 * i)[0] for i in range(40)]
    want = [oracle.compress(a) for a in blocks]
    for _ in range(2):
        res, dst = gpu.encode(blocks)
        for i, w in enumerate(want):
            assert res[i] == len(w) and np.array_equal(dst[i, :res[i]], w), i
        res, dst = gpu.encode(blocks, hc=True)
        for i, a in enumerate(blocks):
            w = oracle.compress(a, hc=True)
            assert res[i] == len(w) and np.array_equal(dst[i, :res[i]], w), i
        _lib.check(_lib.lib().lz4hip_release_workspaces())
        _lib.check(_lib.lib().lz4hip_release_workspaces())            # idempotent


def test_multi_device_host_batches(gpu, oracle):
    """lz4hip_*_batch_host_multi (SURVEY.md 8b `deviceMask`, 8e): block i -> the (i mod N)-th selected device, results in
    global order.  Always with mask 0x1; with two devices (and with mask 0 = all) when the box has them.  Every
    block must be the oracle's bytes whichever device compressed it, and decode back to the input."""
    from lz4net_amd import _lib
    ndev = _lib.lib().lz4hip_device_count()
    masks = [0x1, 0x0]
    if ndev >= 2:
        masks += [0x3, 0x2]
    blocks = _blocks(oracle, sizes=(0, 1, 13, 300, 4096, 65536), seeds=(5,))
    blocks = blocks + [oracle.gen(2, 9, i, 1)[0] for i in range(37)]
    want = [oracle.compress(a) for a in blocks]
    want_hc = [oracle.compress(a, hc=True) for a in blocks]
    for mask in masks:
        for hc, ref in ((False, want), (True, want_hc)):
            res, dst = gpu.encode(blocks, hc=hc, device_mask=mask)
            for i, w in enumerate(ref):
                assert res[i] == len(w) and np.array_equal(dst[i, :res[i]], w), (mask, hc, i)
        used, back = gpu.decode(want, [a.size for a in blocks], known=True, device_mask=mask)
        for i, a in enumerate(blocks):
            assert used[i] == len(want[i]) and np.array_equal(back[i, :a.size], a), (mask, i)
            assert (back[i, a.size:] == 0xA5).all()
        produced, back = gpu.decode(want, [a.size + 3 for a in blocks], known=False, device_mask=mask)
        for i, a in enumerate(blocks):
            assert produced[i] == a.size and np.array_equal(back[i, :a.size], a), (mask, i)
    # a mask that selects nothing visible is an argument error, not a silent fallback
    import ctypes as C
    src, sl = gpu.pack(want)
    caps = np.array([a.size for a in blocks], np.int32)
    dstb = np.zeros((len(want), 65536 + 64), np.uint8)
    resb = np.zeros(len(want), np.int32)
    b = gpu._batch(src, sl, dstb, caps, resb)
    assert _lib.lib().lz4hip_decode_batch_host_multi(C.byref(b), 1, 1 << 63) == _lib.E_ARGUMENT


def _rss_kib():
    for ln in open("/proc/self/status"):
        if ln.startswith("VmRSS:"):
            return int(ln.split()[1])
    return 0


def test_multi_device_threaded_path_does_not_leak(gpu, oracle):
    """The threaded path of lz4hip_*_batch_host_multi (N >= 2 device workers) on whatever devices the box has: the
    "logical_devices" knob runs 3 workers over the visible devices (wrapping around), each with its own staging context.
    The workers are persistent, so 20 calls in a row must leave the free device memory and the process' resident set where
    the second call left them (round 2 started fresh threads per call and abandoned one staging set per device per call:
    device images + 8 pinned buffers + 6 streams + 13 events)."""
    import torch
    from lz4net_amd import _lib
    blocks = [oracle.gen(2, 9, i, 1)[0] for i in range(96)] + [oracle.gen(3, 9, i, 1, 3000 + 41 * i)[0] for i in range(27)]
    want = [oracle.compress(a) for a in blocks]
    sizes = [a.size for a in blocks]
    with _lib.tuning(logical_devices=3):
        def one_round():
            res, dst = gpu.encode(blocks, device_mask=0)
            for i, w in enumerate(want):
                assert res[i] == len(w) and np.array_equal(dst[i, :res[i]], w), i
            used, back = gpu.decode(want, sizes, known=True, device_mask=0)
            for i, a in enumerate(blocks):
                assert used[i] == len(want[i]) and np.array_equal(back[i, :a.size], a), i
        one_round()
        one_round()
        torch.cuda.synchronize()
        free0, rss0 = torch.cuda.mem_get_info()[0], _rss_kib()
        for _ in range(20):
            one_round()
        torch.cuda.synchronize()
        free1, rss1 = torch.cuda.mem_get_info()[0], _rss_kib()
        # one abandoned staging set of this batch is > 16 MiB of device memory and > 16 MiB of pinned host memory
        assert free0 - free1 < (8 << 20), ("device memory shrank", free0 - free1)
        assert rss1 - rss0 < (64 << 10), ("resident set grew (KiB)", rss1 - rss0)
        # the workers' staging is given back on request, and they work again afterwards
        before = torch.cuda.mem_get_info()[0]
        _lib.check(_lib.lib().lz4hip_release_workspaces())
        assert torch.cuda.mem_get_info()[0] > before
        one_round()
        # several caller threads at once share the workers (queueing on them), results stay per caller
        import threading
        errors = []

        def caller():
            try:
                one_round()
            except Exception as e:      # noqa: BLE001
                errors.append(repr(e))
        ts = [threading.Thread(target=caller) for _ in range(3)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors
    _lib.check(_lib.lib().lz4hip_release_workspaces())


def test_multi_device_eight_logical_devices_16384_blocks(gpu, oracle):
    """BASELINE configs[4] shape in miniature, through the C ABI's own sharding (lz4hip_*_batch_host_multi): 16 384 blocks of
    64 KiB round-robin over EIGHT device workers ("logical_devices" = 8 over whatever devices the box has).  Every result in
    global block order, sampled payloads against the oracle, the whole batch round-trips; the wall-clock rates are printed
    (on a one-GPU box they are a code-path check -- eight workers share one device -- not a scaling measurement)."""
    import ctypes as C
    import time
    import torch
    from lz4net_amd import _lib, batch
    n = 16384
    raw_h = batch.synth(2, 515, 0, n).cpu().numpy()
    comp_h = np.zeros((n, batch.BOUND_STRIDE), np.uint8)
    lens = np.full(n, batch.BLOCK, np.int32)
    caps = np.full(n, batch.BOUND, np.int32)
    clen = np.zeros(n, np.int32)
    eb = _lib.Batch(src=raw_h.ctypes.data, src_off=None, src_stride=raw_h.strides[0], src_len=lens.ctypes.data,
                    dst=comp_h.ctypes.data, dst_off=None, dst_stride=comp_h.strides[0], dst_cap=caps.ctypes.data,
                    dst_cap_all=0, src_len_all=0, result=clen.ctypes.data, n_blocks=n)
    back_h = np.zeros_like(raw_h)
    res = np.zeros(n, np.int32)
    db = _lib.Batch(src=comp_h.ctypes.data, src_off=None, src_stride=comp_h.strides[0], src_len=clen.ctypes.data,
                    dst=back_h.ctypes.data, dst_off=None, dst_stride=back_h.strides[0], dst_cap=lens.ctypes.data,
                    dst_cap_all=0, src_len_all=0, result=res.ctypes.data, n_blocks=n)
    with _lib.tuning(logical_devices=8):
        t0 = time.perf_counter()
        _lib.check(_lib.lib().lz4hip_encode_batch_host_multi(C.byref(eb), 0, C.c_uint64(0)))
        t1 = time.perf_counter()
        _lib.check(_lib.lib().lz4hip_decode_batch_host_multi(C.byref(db), 1, C.c_uint64(0)))
        t2 = time.perf_counter()
    assert (clen > 0).all() and (res == clen).all()
    assert np.array_equal(back_h, raw_h)
    for i in list(range(0, n, 1021)) + [n - 1]:
        w = oracle.compress(raw_h[i])
        assert clen[i] == len(w) and np.array_equal(comp_h[i, :len(w)], w), i
    print(f"8 logical devices over {torch.cuda.device_count()} physical, {n} blocks: encode {n * 65536 / (t1 - t0) / 1e9:.2f} GB/s, "
          f"decode {n * 65536 / (t2 - t1) / 1e9:.2f} GB/s wall clock (first call of the workers, staging allocation included)")
    _lib.check(_lib.lib().lz4hip_release_workspaces())


def test_hc_host_batch_blocks_over_64k(gpu, oracle):
    """LZ4HC through the host-pointer entry point with blocks above 64 KiB (32-bit heads, the kernels with the insert loop):
    the slice size of the staging pipeline is bounded in BYTES whatever the rows are (the 16384-block slice hint
    of the LZ4HC path was applied to rows of any size).  40 blocks of 70 000 .. 600 000 bytes; every block against the oracle."""
    rng = np.random.default_rng(77)
    blocks = []
    for i in range(40):
        sz = int(rng.choice([70000, 65537, 131072, 200000, 600000]))
        blocks.append(oracle.gen(2 if i % 2 else 3, 71, i * 16, (sz + 65535) // 65536).reshape(-1)[:sz].copy())
    res, dst = gpu.encode(blocks, hc=True)
    for i, a in enumerate(blocks):
        w = oracle.compress(a, hc=True)
        assert res[i] == len(w), (i, a.size, res[i], len(w))
        assert np.array_equal(dst[i, :res[i]], w), (i, a.size)


def test_hc_sub_chunks_identical(gpu, oracle):
    """The pipelined LZ4HC lane launch (sub-chunks whose table builders and lane kernels overlap on separate streams) must
    produce the bytes of the one-after-the-other launch: 8192 blocks, 1 / 2 / 3 sub-chunks, sampled blocks against the oracle."""
    import torch
    from lz4net_amd import _lib, batch
    n = 8192
    raw = batch.synth(2, 808, 0, n)
    host = raw.cpu().numpy()
    outs = []
    for subs in (1, 2, 3):
        comp = torch.full((n, batch.BOUND_STRIDE), 0xA5, dtype=torch.uint8, device="cuda")
        with ForcedMapping("LZ4HIP_HC", "lane"), _lib.tuning(hc_sub_chunks=subs):
            clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND, hc=True)
            torch.cuda.synchronize()
        outs.append((clen.cpu().numpy(), batch.checksum(comp, clen).cpu().numpy()))
        lens = outs[-1][0]
        for i in (0, 1, n // 2 - 1, n // 2, n - 1):
            w = oracle.compress(host[i], hc=True)
            assert lens[i] == len(w) and np.array_equal(comp[i, :len(w)].cpu().numpy(), w), (subs, i)
    for k in (1, 2):
        assert np.array_equal(outs[0][0], outs[k][0]) and np.array_equal(outs[0][1], outs[k][1]), k


def test_hc_lane_slab_reuse(gpu, oracle):
    """LZ4HC lane kernel with ONE wavefront in the grid ("hc_groups" = 1): its 64 lanes encode 320 blocks, five each on
    average, on slabs that are never re-initialised beyond the heads and chain[0] -- blocks of 64 KiB, 70 000 bytes
    (32-bit heads, a different slab layout) and 300 .. 30 000 bytes in random order, fuzzer-style and record-like, some
    with long repeats.  EVERY block is compared with the oracle."""
    from lz4net_amd import _lib
    rng = np.random.default_rng(43)
    blocks = []
    for i in range(320):
        k = int(rng.integers(0, 10))
        sz = 65536 if k < 2 else 70000 if k == 2 else 65536 - int(rng.integers(1, 2000)) if k == 3 else int(rng.integers(300, 30000))
        row = oracle.gen(2 if i % 3 else 3, 61, i, (sz + 65535) // 65536).reshape(-1)[:sz].copy()
        if i % 6 == 0:
            row[sz // 2:] = row[:sz - sz // 2]
        if i % 11 == 0:
            row[: sz // 4] = 9
        blocks.append(row)
    with ForcedMapping("LZ4HIP_HC", "lane"), _lib.tuning(hc_groups=1):
        res, dst = gpu.encode(blocks, hc=True)
    for i, a in enumerate(blocks):
        w = oracle.compress(a, hc=True)
        assert res[i] == len(w), (i, a.size, res[i], len(w))
        assert np.array_equal(dst[i, :res[i]], w), (i, a.size)


def _nat_blocks(oracle, n, seed, big_every=40):
    """Blocks <= 64 KiB that stress the exactness arguments of lz4hip_hc_nat.hpp / lz4hip_hc_lcp.hpp: short-period runs, tiny alphabets, copies of
    earlier content, runs with single disturbed bytes, plus fuzzer-style and record-like rows."""
    rng = np.random.default_rng(seed)
    blocks = []
    for i in range(n):
        mode = i % 6
        sz = int(rng.integers(13, 6000)) if i % big_every else 65536 - int(rng.integers(0, 3000))
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
            row = oracle.gen(2, 300 + seed, i, 1).reshape(-1)[:sz].copy()
            row[sz // 2:] = row[:sz - sz // 2]
        elif mode == 3:
            row = oracle.gen(3, 300 + seed, i, 1).reshape(-1)[:sz].copy()
            row[: sz // 3] = row[0]
        elif mode == 4:
            row = np.full(sz, int(rng.integers(0, 256)), np.uint8)
            for _ in range(int(rng.integers(0, 40))):
                row[int(rng.integers(0, sz))] = int(rng.integers(0, 256))
        else:
            row = oracle.gen(2, 400 + seed, i, 1).reshape(-1)[:sz].copy()
        blocks.append(row)
    return blocks


def test_hc_lane_small_blocks_many_per_lane(gpu, oracle):
    """The LZ4HC lane kernel for blocks <= 64 KiB (lz4hip_hc_lcp.hpp: chains and shared lengths of the whole chunk built first
    by hc_nat_chain_kernel + hc_lcp_fill_kernel, then the state machine without an insert loop) with ONE wavefront in the
    grid: 640 blocks, ten per lane.  EVERY block is compared with the oracle."""
    from lz4net_amd import _lib
    blocks = _nat_blocks(oracle, 640, 5)
    with ForcedMapping("LZ4HIP_HC", "lane"), _lib.tuning(hc_groups=1):
        res, dst = gpu.encode(blocks, hc=True)
    for i, a in enumerate(blocks):
        w = oracle.compress(a, hc=True)
        assert res[i] == len(w), (i, a.size, res[i], len(w))
        assert np.array_equal(dst[i, :res[i]], w), (i, a.size)


def test_hc_precomputed_chains_several_chunks(gpu, oracle):
    """lz4hip_hc_lcp.hpp over a batch larger than its chunk (4096 tables with one wavefront in the grid): 9000 short blocks in ONE
    device batch = three rounds of table builders + lane kernel on the same tables."""
    from lz4net_amd import _lib
    blocks = _nat_blocks(oracle, 9000, 9, big_every=3000)
    blocks = [b[:2500] for b in blocks]
    with ForcedMapping("LZ4HIP_HC", "lane"), _lib.tuning(hc_groups=1, host_slices=1):
        res, dst = gpu.encode(blocks, hc=True)
    for i, a in enumerate(blocks):
        w = oracle.compress(a, hc=True)
        assert res[i] == len(w), (i, a.size, res[i], len(w))
        assert np.array_equal(dst[i, :res[i]], w), (i, a.size)


def test_hc_host_batch_takes_the_lane_mapping_from_4096_blocks(gpu, oracle):
    """Default dispatch, nothing forced: a host-pointer LZ4HC batch of 4200 blocks <= 64 KiB goes to the device in ONE slice
    (LZ4HC slices are 16384 blocks) and is encoded by the lane mapping (table builders + lz4hip_hc_lcp.hpp); 4000 blocks by the
    wavefront mapping.  Every block is compared with the oracle."""
    from lz4net_amd import _lib
    blocks = [b[:700] for b in _nat_blocks(oracle, 4200, 13, big_every=100000)]
    want = [oracle.compress(a, hc=True) for a in blocks]
    for n, lane_expected in ((4200, True), (4000, False)):
        before = _lib.dispatch_counts()
        res, dst = gpu.encode(blocks[:n], hc=True)
        after = _lib.dispatch_counts()
        assert (after[_lib.K_HC_LANE] > before[_lib.K_HC_LANE]) == lane_expected, (n, before, after)
        assert (after[_lib.K_HC_WAVE] > before[_lib.K_HC_WAVE]) == (not lane_expected), (n, before, after)
        for i in range(n):
            assert res[i] == len(want[i]) and np.array_equal(dst[i, :res[i]], want[i]), (n, i)
