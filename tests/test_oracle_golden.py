"""CPU oracle vs the committed golden vectors (generated from the reference's own C by
tests/golden/make_golden.py).  Runs everywhere, including boxes without /root/reference."""
import hashlib
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vectors.json")))


def sha(a):
    return hashlib.sha256(bytes(a)).hexdigest()


def test_compress_bound(oracle):
    # original/lz4.h:85-86 == src/LZ4ps/LZ4Codec.cs:142-145
    from oracle.oracle import compress_bound
    for n, want in GOLD["compress_bound"].items():
        assert compress_bound(int(n)) == want
        assert oracle.lib.lz4o_compress_bound(int(n)) == want
    assert compress_bound(65536) == 65809


@pytest.mark.parametrize("idx", range(len(GOLD["synth"])))
def test_synth_vectors(oracle, idx):
    e = GOLD["synth"][idx]
    a = oracle.gen(e["dist"], e["seed"], e["block"], 1, max(e["n"], 1))[0][:e["n"]]
    assert sha(a) == e["input_sha256"], "synthetic generator drifted from the golden byte stream"
    f = oracle.compress(a)
    assert (len(f), sha(f)) == (e["fast_len"], e["fast_sha256"])
    h = oracle.compress(a, hc=True)
    assert (len(h), sha(h)) == (e["hc_len"], e["hc_sha256"])
    assert np.array_equal(oracle.uncompress(f, a.size), a)
    assert np.array_equal(oracle.uncompress(h, a.size), a)
    ret, out = oracle.uncompress_unknown_raw(h, len(h), a.size + 3)
    assert ret == a.size and np.array_equal(out[:ret], a)


@pytest.mark.parametrize("e", GOLD["inline"], ids=[e["name"] for e in GOLD["inline"]])
def test_inline_vectors(oracle, e):
    if "input_hex" in e:
        a = np.frombuffer(bytes.fromhex(e["input_hex"]), dtype=np.uint8)
    elif e["name"] == "zeros_65536":
        a = np.zeros(65536, dtype=np.uint8)
    else:
        a = np.array([((i * 2654435761) & 0xFFFFFFFF) >> 24 for i in range(65536)], dtype=np.uint8)
    assert sha(a) == e["input_sha256"]
    f, h = oracle.compress(a), oracle.compress(a, hc=True)
    assert (len(f), sha(f)) == (e["fast_len"], e["fast_sha256"])
    assert (len(h), sha(h)) == (e["hc_len"], e["hc_sha256"])
    if "fast_hex" in e:
        assert bytes(f).hex() == e["fast_hex"] and bytes(h).hex() == e["hc_hex"]
    # limited output: exact fits, one byte less returns 0 (original/fuzzer.c:212-227)
    assert oracle.compress_raw(a, len(f))[0] == e["fast_cap_exact"]
    assert oracle.compress_raw(a, len(f) - 1)[0] == e["fast_cap_minus1"] == 0
    assert oracle.compress_raw(a, len(h), hc=True)[0] == e["hc_cap_exact"]
    assert oracle.compress_raw(a, len(h) - 1, hc=True)[0] == e["hc_cap_minus1"] == 0
    # decoder return codes (original/fuzzer.c:185-210)
    assert oracle.uncompress_raw(f, a.size)[0] == e["dec_known"]
    assert oracle.uncompress_raw(f, a.size + 1)[0] == e["dec_known_plus1"]
    if a.size:
        assert oracle.uncompress_raw(f, a.size - 1)[0] == e["dec_known_minus1"]
        assert oracle.uncompress_unknown_raw(f, len(f), a.size - 1)[0] == e["dec_unknown_out_minus1"]
    assert oracle.uncompress_unknown_raw(f, len(f), a.size)[0] == e["dec_unknown_exact"]
    assert oracle.uncompress_unknown_raw(f, len(f), a.size + 1)[0] == e["dec_unknown_room"]
    assert oracle.uncompress_unknown_raw(f, len(f) - 1, a.size)[0] == e["dec_unknown_in_minus1"]
    assert oracle.uncompress_unknown_raw(f, len(f) + 1, a.size)[0] == e["dec_unknown_in_plus1"]


def test_zero_block_known_answer(oracle):
    # SURVEY.md 8c KAT (1): hand-derived from original/lz4.c:631-767
    f = oracle.compress(np.zeros(65536, dtype=np.uint8))
    want = bytes([0x1F, 0x00, 0x01, 0x00]) + b"\xff" * 256 + bytes([0xE7, 0x50, 0, 0, 0, 0, 0])
    assert bytes(f) == want and len(f) == 267
