"""The lane encoder's table slab: candidate placements built and measured by the library (knob encoder_slab_tries) against the first, unmeasured one.
One fresh process per call: python tools/enc_slab_calibration.py <tries> [blocks]   (the batch buffers -- 192 GB at 2^20 blocks -- are allocated first,
as in bench.py)."""
import sys
import time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib

tries = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
raw = batch.synth(2, 20260925, 0, n)
comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
back = torch.empty_like(raw)
_lib.tuning_set("encoder_slab_tries", tries)
torch.cuda.synchronize()
t = time.perf_counter()
clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
torch.cuda.synchronize()
first = time.perf_counter() - t
ts = []
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); batch.encode(raw, batch.BLOCK, comp, batch.BOUND); b.record(); b.synchronize()
    ts.append(a.elapsed_time(b))
batch.decode(comp, clen, back, batch.BLOCK)
ok = bool(torch.equal(back, raw))
print("encoder_slab_tries=%d: encode %s GB/s (first call incl. slab set-up %.0f ms); slab placement measured %.2f G steps/s, %d candidate(s) built; round trip ok=%s" % (
    tries, " / ".join("%.2f" % (n * 65536 / x / 1e6) for x in ts), first * 1e3, _lib.tuning_get("encoder_slab_rate") / 1000.0,
    _lib.tuning_get("encoder_slab_tried"), ok), flush=True)
