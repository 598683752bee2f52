"""Generates tests/golden/vectors.json from the REFERENCE's own C (oracle/_ref/libref_lz4.so, built in
place from /root/reference/original by oracle/Makefile).  Run in the build container only:

    python tests/golden/make_golden.py

The reference ships no golden vectors, so these are the known-answer tests that pin
both the CPU oracle (-m "not gpu") and the HIP path (-m gpu) on boxes where /root/reference is absent.
Inputs are described by (distribution, seed, block index, length) of oracle/synth.c -- whose byte
streams are pinned here too via sha256 -- or given inline as hex.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, Reference, compress_bound  # noqa: E402

LOREM = (b"Lorem ipsum dolor sit amet, consectetur adipisicing elit, sed do eiusmod tempor incididunt ut "
         b"labore et dolore magna aliqua. Ut enim ad minim veniam, quis nostrud exercitation ullamco laboris "
         b"nisi ut aliquip ex ea commodo consequat. Duis aute irure dolor in reprehenderit in voluptate velit "
         b"esse cillum dolore eu fugiat nulla pariatur. Excepteur sint occaecat cupidatat non proident, sunt "
         b"in culpa qui officia deserunt mollit anim id est laborum.")


def sha(a) -> str:
    return hashlib.sha256(bytes(a)).hexdigest()


def main():
    o, r = Oracle(), Reference()
    synth = []
    for dist in range(4):
        for seed, block in ((1, 0), (1, 7), (42, 123456)):
            for n in (0, 1, 12, 13, 64, 1000, 4096, 32768, 65535, 65536, 65546, 65547, 100000):
                a = o.gen(dist, seed, block, 1, max(n, 1))[0][:n]
                f = r.compress(a, hc=False)
                h = r.compress(a, hc=True)
                synth.append(dict(dist=dist, seed=seed, block=block, n=n, input_sha256=sha(a),
                                  fast_len=len(f), fast_sha256=sha(f), hc_len=len(h), hc_sha256=sha(h)))
    inline = []
    mul = np.array([((i * 2654435761) & 0xFFFFFFFF) >> 24 for i in range(65536)], dtype=np.uint8)
    cases = {
        "abcdefghijkl": np.frombuffer(b"abcdefghijkl", dtype=np.uint8),
        "lorem_x1": np.frombuffer(LOREM, dtype=np.uint8),
        "lorem_x5": np.frombuffer(LOREM * 5, dtype=np.uint8),
        "zeros_65536": np.zeros(65536, dtype=np.uint8),
        "ramp_mul_65536": mul,
        "single_byte": np.frombuffer(b"x", dtype=np.uint8),
    }
    for name, a in cases.items():
        f = r.compress(a, hc=False)
        h = r.compress(a, hc=True)
        e = dict(name=name, n=int(a.size), input_sha256=sha(a), fast_len=len(f), fast_sha256=sha(f),
                 hc_len=len(h), hc_sha256=sha(h))
        if a.size <= 4096 or len(f) <= 512:
            e["fast_hex"] = bytes(f).hex()
            e["hc_hex"] = bytes(h).hex()
        if a.size <= 4096:
            e["input_hex"] = bytes(a).hex()
        # limited-output behaviour (original/fuzzer.c:212-224): exact size OK, one less => 0
        e["fast_cap_exact"] = r.compress_raw(a, len(f))[0]
        e["fast_cap_minus1"] = r.compress_raw(a, len(f) - 1)[0]
        e["hc_cap_exact"] = r.compress_raw(a, len(h), hc=True)[0]
        e["hc_cap_minus1"] = r.compress_raw(a, len(h) - 1, hc=True)[0]
        # decoder return codes on the fast stream
        e["dec_known"] = r.uncompress_raw(f, a.size)[0]
        e["dec_known_minus1"] = r.uncompress_raw(f, a.size - 1)[0] if a.size else None
        e["dec_known_plus1"] = r.uncompress_raw(f, a.size + 1)[0]
        e["dec_unknown_exact"] = r.uncompress_unknown_raw(f, len(f), a.size)[0]
        e["dec_unknown_room"] = r.uncompress_unknown_raw(f, len(f), a.size + 1)[0]
        e["dec_unknown_out_minus1"] = r.uncompress_unknown_raw(f, len(f), a.size - 1)[0] if a.size else None
        e["dec_unknown_in_minus1"] = r.uncompress_unknown_raw(f, len(f) - 1, a.size)[0]
        e["dec_unknown_in_plus1"] = r.uncompress_unknown_raw(f, len(f) + 1, a.size)[0]
        inline.append(e)
    out = dict(generator="tests/golden/make_golden.py", source="oracle/_ref/libref_lz4.so (original/lz4.c, lz4hc.c; "
               "-DLZ4_ARCH64=1 -DLZ4_MK_OPT)", synth=synth, inline=inline,
               compress_bound={str(n): compress_bound(n) for n in (0, 1, 254, 255, 65536, 1 << 20)})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(synth), "synthetic +", len(inline), "inline vectors")


if __name__ == "__main__":
    main()
