"""Fast-encode rate vs the row stride of the INPUT batch (the bench's rows are 64 KiB apart -- a power of two) and vs re-allocating the
batch buffers: is the bimodal rate (44-46 vs 53-54 GB/s) channel aliasing of the lanes' input / output streams?"""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
for rep in range(2):
    for pad_in, pad_out in ((0, 0), (256, 0), (4160, 0), (0, 4096 - 288), (0, 0)):
        raw = batch.synth(2, 20260925, 0, n, stride=batch.BLOCK + pad_in)
        comp = torch.empty((n, batch.BOUND_STRIDE + pad_out), dtype=torch.uint8, device="cuda")
        ts = []
        for _ in range(2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
        print("rep %d input stride %d, output stride %d (raw @ 0x%x, comp @ 0x%x): encode %.2f GB/s" % (rep, raw.stride(0), comp.stride(0), raw.data_ptr(), comp.data_ptr(), n * 65536 / min(ts) / 1e6), flush=True)
        del raw, comp
        torch.cuda.empty_cache()
