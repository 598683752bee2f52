"""LZ4HC lane mapping: generation 1 (one loop nest per lane) vs generation 2 (convergent state machine), device-resident batch,
per residency; every block's compressed length + checksum compared between the two (and a sample with the CPU oracle).
usage: python tools/hc_gen_ab.py [blocks] ["gen:waves_per_cu[:ctrl_every[:ctrl_lanes]],..."] [dists]
(generations: 1 lz4hip_hc_lane.hpp and 2 lz4hip_hc_conv.hpp<u16> only in -DLZ4HIP_TUNING_BUILD libraries, 3 lz4hip_hc_nat.hpp, 4 lz4hip_hc_lcp.hpp)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lz4net_amd import batch, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
cfgs = [tuple(int(x) for x in c.split(":")) for c in (sys.argv[2] if len(sys.argv) > 2 else "1:16,2:16,2:8,2:4").split(",")]
dists = [int(d) for d in (sys.argv[3] if len(sys.argv) > 3 else "2,3").split(",")]
_lib.tuning_set("hc", "lane")
for dist in dists:
    raw = batch.synth(dist, 20260925, 0, n)
    comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
    clen = torch.empty(n, dtype=torch.int32, device="cuda")
    ref = None
    for cfg in cfgs:
        gen, wpc = cfg[0], cfg[1]
        every = cfg[2] if len(cfg) > 2 else 0          # generation 4: control-flow batching interval / lanes (0 = default)
        lanes = cfg[3] if len(cfg) > 3 else 0
        _lib.tuning_set("hc_gen", gen); _lib.tuning_set("hc_waves_per_cu", wpc)
        _lib.tuning_set("hc_ctrl_every", every); _lib.tuning_set("hc_ctrl_lanes", lanes)
        batch.encode(raw[:16384], batch.BLOCK, comp[:16384], batch.BOUND, hc=True, result=clen[:16384])   # workspace for this residency
        torch.cuda.synchronize()
        ts = []
        for _ in range(2):                # (the first pass of a residency also allocates its workspace inside the launch call)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); batch.encode(raw, batch.BLOCK, comp, batch.BOUND, hc=True, result=clen); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
        t = min(ts)
        passes = ", ".join("%.0f" % x for x in ts)
        sig = (batch.checksum(comp, clen).cpu().numpy().copy(), clen.cpu().numpy().copy())
        same = None if ref is None else bool((sig[0] == ref[0]).all() and (sig[1] == ref[1]).all())
        if ref is None:
            ref = sig
        print(f"dist={dist} blocks={n} hc_gen={gen} waves/CU={wpc} ctrl={every}/{lanes}: {n * 65536 / t / 1e6:7.3f} GB/s  {t:9.1f} ms (passes: {passes})  ratio {float(clen.double().sum()) / (n * 65536):.4f}  same bytes as first config: {same}", flush=True)
    del raw, comp
    torch.cuda.empty_cache()
