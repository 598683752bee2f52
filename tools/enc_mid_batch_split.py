"""Mid-size fast-encode batches (what a host-pointer call's slices and callers with a few thousand blocks hit): the wavefront mapping alone, the lane
mapping alone, and BOTH at once on disjoint parts of the batch (two streams).  usage: python tools/enc_mid_batch_split.py [blocks] [dist]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dist = int(sys.argv[2]) if len(sys.argv) > 2 else 2
raw = batch.synth(dist, 20260925, 0, n)
comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
ref = torch.empty_like(comp)
res = torch.empty(n, dtype=torch.int32, device="cuda")
_lib.tuning_set("encoder", "wave")
want = batch.encode(raw, batch.BLOCK, ref, batch.BOUND).clone()
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def run(k):          # blocks [0, k) on the wavefront mapping, [k, n) on the lane mapping
    comp.zero_(); res.zero_()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    sa.wait_event(a); sb.wait_event(a)
    if k > 0:
        _lib.tuning_set("encoder", "wave")
        with torch.cuda.stream(sa):
            batch.encode(raw[:k], batch.BLOCK, comp[:k], batch.BOUND, result=res[:k])
    if k < n:
        _lib.tuning_set("encoder", "lane")
        with torch.cuda.stream(sb):
            batch.encode(raw[k:], batch.BLOCK, comp[k:], batch.BOUND, result=res[k:])
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    ea.record(sa); eb.record(sb)
    torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
    b.record(); b.synchronize()
    ok = bool((res == want).all()) and batch.count_mismatches(comp, ref, want) == 0
    return a.elapsed_time(b), ok


for frac in (1.0, 0.0, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3):
    k = int(n * frac) // 64 * 64
    run(k)
    t, ok = min(run(k) for _ in range(3))
    print(f"dist {dist} blocks {n}: {k:6d} on the wavefront mapping, {n - k:6d} on the lane mapping: {t:8.2f} ms  {n * 65536 / t / 1e6:7.2f} GB/s  bytes equal {ok}", flush=True)
