"""Same-box A/B of lane-decoder configurations selected through the tuning knobs (decoder_gen, decoder_ring): the workload of
each distribution is built ONCE (device-side synth + bit-exact fast encode), then every configuration decodes it.
usage: python tools/ab_decoder_knobs.py [blocks] ["gen:ring,gen:ring,..."] [dists]      (needs a library built with -DLZ4HIP_TUNING_BUILD
for ring sizes other than the default)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lz4net_amd import batch, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
cfgs = [tuple(int(x) for x in c.split(":")) for c in (sys.argv[2] if len(sys.argv) > 2 else "2:128,3:128").split(",")]
dists = [int(d) for d in (sys.argv[3] if len(sys.argv) > 3 else "2,3").split(",")]
steps = int(os.environ.get("STEPS", "3"))
for dist in dists:
    raw = batch.synth(dist, 20260925, 0, n)
    comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
    back = torch.empty_like(raw)
    used = torch.empty(n, dtype=torch.int32, device="cuda")
    alg = n * batch.BLOCK + int(clen.to(torch.int64).sum().item()) + 8 * n
    _lib.tuning_set("decoder", "lane")
    for gen, ring in cfgs:
        _lib.tuning_set("decoder_gen", gen)
        _lib.tuning_set("decoder_ring", ring if gen >= 3 else 0)
        try:
            back.zero_()
            batch.decode(comp, clen, back, batch.BLOCK, result=used)
            torch.cuda.synchronize()
        except Exception as e:
            print(f"dist={dist} gen={gen} ring={ring}: {e!r}", flush=True)
            continue
        ok = bool((used == clen).all()) and batch.count_mismatches(raw, back, batch.BLOCK) == 0
        ms = []
        for _ in range(steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); batch.decode(comp, clen, back, batch.BLOCK, result=used); b.record(); b.synchronize()
            ms.append(a.elapsed_time(b))
        t = min(ms)
        print(f"dist={dist} blocks={n} gen={gen} ring={ring}: {n * batch.BLOCK / t / 1e6:8.1f} GB/s  {t:8.3f} ms  frac {alg / t / 1e6 / 8000:.4f}  ok={ok}", flush=True)
    del raw, comp, back
    torch.cuda.empty_cache()
