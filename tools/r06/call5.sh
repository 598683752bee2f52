#!/bin/bash
# round 6, call 5: do the two grids of the shared hand-over overlap?  kernel trace of a 2^18-block D2 fast encode; plus the wavefront decoder's s_memtime sections
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call5; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/enc_once.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from lz4net_amd import batch, _lib
n = 262144
raw = batch.synth(2, 20260925, 0, n)
comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
for rep in range(2):
    clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
    torch.cuda.synchronize()
print("ok", int(clen.sum()))
PY
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python /tmp/enc_once.py > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r06_call5/encoder_shared_kernel_trace.txt
import csv, glob
f = sorted(glob.glob("gpurun_out/r06_call5/trace/**/*kernel_trace.csv", recursive=True))
rows = list(csv.DictReader(open(f[-1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    name = r["Kernel_Name"]
    if "encode" in name or "slab" in name:
        print("%-70s start %10.3f ms  end %10.3f ms  dur %9.3f ms" % (name[:70], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilz4net_amd/csrc tools/dec_wave_sections.hip -o /tmp/dec_wave_sections 2>/dev/null
for d in 2 3; do for n in 1024 4096; do timeout 120 /tmp/dec_wave_sections $n $d; done; done 2>&1 | tee $O/decoder_wave_sections.txt
