"""Is the first large LZ4HC batch slower than the second (workspace allocation inside the timed region)?"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from lz4net_amd import batch
n = 1 << 18
raw = batch.synth(2, 1, 0, n)
comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
batch.encode(raw[:64], batch.BLOCK, comp[:64], batch.BOUND, hc=True)
torch.cuda.synchronize()
for i in range(3):
    t0 = time.perf_counter()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND, hc=True); b.record(); b.synchronize()
    print("call %d: events %.1f ms, wall %.1f ms -> %.2f GB/s" % (i, a.elapsed_time(b), (time.perf_counter() - t0) * 1e3, n * 65536 / a.elapsed_time(b) / 1e6), flush=True)
