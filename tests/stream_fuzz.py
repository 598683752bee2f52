"""Generator of arbitrary LZ4 block streams (not the output of any particular encoder): sequences with random
literal-run lengths, match lengths and offsets, including the corners an encoder rarely or never produces --
zero-length literal runs in a row, offsets of 1..8 and around the decoders' internal thresholds, length bytes of
exactly 255 / 254 / 0 after a saturated nibble, matches that start inside their own output (offset < length),
offset 0, and streams that end too early or too late.  The oracle decides what the right answer is (bytes and
return code, for both decoders); the kernels must agree.  Test infrastructure only."""
import numpy as np


def _put_len(out, n):
    while n >= 255:
        out.append(255)
        n -= 255
    out.append(n)


def make_stream(rng, target, tail_kind=0):
    """Returns (compressed bytes, expected output bytes if the stream is well formed else None)."""
    out, raw = bytearray(), bytearray()
    while True:
        remaining = target - len(raw)
        style = rng.integers(0, 10)
        ll = int(rng.choice([0, 0, 0, 1, 2, 3, 7, 11, 12, 14, 15, 16, 30, 269, 270, 271, 600])) if style < 8 else int(rng.integers(0, 40))
        ml = int(rng.choice([4, 4, 5, 6, 7, 8, 12, 15, 16, 17, 18, 19, 20, 32, 33, 64, 65, 273, 274, 275, 1000])) if style < 8 else int(rng.integers(4, 80))
        if remaining - ll - ml < 13 or (len(raw) + ll == 0):
            # final literal run: everything that is left (at least 5 bytes by construction)
            run = remaining
            lits = rng.integers(0, 256, run, dtype=np.uint8).tobytes() if rng.integers(0, 3) else bytes([int(rng.integers(0, 4))]) * run
            out.append((min(run, 15) << 4))
            if run >= 15:
                _put_len(out, run - 15)
            out += lits
            raw += lits
            break
        lits = rng.integers(0, 256, ll, dtype=np.uint8).tobytes()
        avail = len(raw) + ll
        off_choices = [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 47, 48, 49, 63, 64, 65, 66, 100, 107, 108, 109, 110, 111, 112, 113, 127, 128, 129, 170, 171, 172, 173, 174, 188, 191, 192, 193,
                       255, 256, 257, 1000, 4031, 4032, 4033, 4094, 4095, 4096, 4097, 4100, 8192, 65535]     # (around every ring / window / burst threshold of the decoders)
        off = int(rng.choice([o for o in off_choices if o <= avail] or [avail])) if style < 9 else int(rng.integers(1, avail + 1))
        out.append((min(ll, 15) << 4) | min(ml - 4, 15))
        if ll >= 15:
            _put_len(out, ll - 15)
        out += lits
        raw += lits
        out += bytes([off & 255, off >> 8])
        if ml - 4 >= 15:
            _put_len(out, ml - 4 - 15)
        start = len(raw) - off
        for i in range(ml):
            raw.append(raw[start + i])
    comp = bytes(out)
    if tail_kind == 1:                      # truncated
        comp = comp[:max(1, len(comp) - int(rng.integers(1, 9)))]
    elif tail_kind == 2:                    # trailing garbage
        comp = comp + bytes(rng.integers(0, 256, int(rng.integers(1, 6)), dtype=np.uint8).tobytes())
    elif tail_kind == 3 and len(comp) > 8:  # one corrupted byte
        b = bytearray(comp)
        b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        comp = bytes(b)
    elif tail_kind == 4 and len(comp) > 8:  # an offset of zero somewhere
        b = bytearray(comp)
        k = int(rng.integers(1, len(b) - 1))
        b[k] = 0; b[k + 1] = 0
        comp = bytes(b)
    return np.frombuffer(comp, dtype=np.uint8).copy(), (np.frombuffer(bytes(raw), dtype=np.uint8).copy() if tail_kind == 0 else None)


def make_zero_offset_stream(rng):
    """A match with offset 0 (its bytes keep whatever the destination holds) directly followed by SHORT sequences whose
    matches read those very bytes back -- the case in which a decoder that mirrors its recent output somewhere (LDS ring)
    must not serve stale mirror bytes.  Returns (stream, output size); the oracle defines the expected bytes."""
    out = bytearray()
    size = 0

    def seq(ll, ml, off):
        nonlocal size
        out.append((min(ll, 15) << 4) | min(ml - 4, 15))
        if ll >= 15:
            _put_len(out, ll - 15)
        out.extend(rng.integers(0, 256, ll, dtype=np.uint8).tobytes())
        out.extend(bytes([off & 255, off >> 8]))
        if ml - 4 >= 15:
            _put_len(out, ml - 4 - 15)
        size += ll + ml

    seq(int(rng.integers(1, 30)), int(rng.integers(4, 19)), 1 + int(rng.integers(0, 1)))       # something before
    for _ in range(int(rng.integers(1, 4))):
        ml0 = int(rng.integers(4, 40))
        seq(int(rng.integers(0, 3)), ml0, 0)                                                   # offset 0
        for _ in range(int(rng.integers(1, 4))):                                               # short matches into its bytes
            ll = int(rng.integers(0, 3))
            seq(ll, int(rng.integers(4, 19)), int(rng.integers(1, ml0 + ll + 1)))
    run = 12 + int(rng.integers(0, 20))
    out.append(min(run, 15) << 4)
    if run >= 15:
        _put_len(out, run - 15)
    out.extend(rng.integers(0, 256, run, dtype=np.uint8).tobytes())
    size += run
    return np.frombuffer(bytes(out), dtype=np.uint8).copy(), size


def is_zero_offset_case(i):
    return i % 10 == 9


def has_zero_offset(c, limit):
    """Does a decoder walking stream `c` (stopping once `limit` output bytes are reached or the stream ends) meet a match
    with offset 0?  For such streams only the return codes are compared with the REFERENCE: the bytes of an offset-0 match
    are whatever the destination held, which in the reference includes the overshoot of its own 8-byte wild copies -- an
    artefact, not a result.  The kernels are compared with the oracle (exact-length copies: a hole keeps the caller's
    bytes), bytes included."""
    c = bytes(c)
    ip = op = 0
    n = len(c)
    while ip < n and op <= limit:
        tok = c[ip]; ip += 1
        ll = tok >> 4
        if ll == 15:
            while ip < n:
                b = c[ip]; ip += 1; ll += b
                if b != 255:
                    break
        ip += ll; op += ll
        if ip + 2 > n or op >= limit:
            return False
        if c[ip] == 0 and c[ip + 1] == 0:
            return True
        ip += 2
        ml = tok & 15
        if ml == 15:
            while ip < n:
                b = c[ip]; ip += 1; ml += b
                if b != 255:
                    break
        op += ml + 4
    return False


def cases(seed, count, max_size=6000):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        if is_zero_offset_case(i):
            c, size = make_zero_offset_stream(rng)
            out.append(((c, None), size))
            continue
        target = int(rng.choice([13, 14, 20, 64, 65, 200, 1000, 3000, max_size])) + int(rng.integers(0, 50))
        out.append((make_stream(rng, target, tail_kind=0 if i % 3 else int(rng.integers(0, 5))), target))
    return out
