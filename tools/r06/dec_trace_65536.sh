cd /tmp && export TMPDIR=/tmp
cat > /tmp/dec_once.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from lz4net_amd import batch, _lib
n = 65536
raw = batch.synth(2, 20260925, 0, n)
comp = torch.empty((n, batch.BOUND_STRIDE), dtype=torch.uint8, device="cuda")
clen = batch.encode(raw, batch.BLOCK, comp, batch.BOUND)
back = torch.empty_like(raw)
for rep in range(3):
    batch.decode(comp, clen, back, batch.BLOCK)
    torch.cuda.synchronize()
print("ok")
PY
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_dec -- python /tmp/dec_once.py > /tmp/trace_dec.log 2>&1
python - <<'PY'
import csv, glob
f = sorted(glob.glob("/tmp/trace_dec/**/*kernel_trace.csv", recursive=True))
rows = list(csv.DictReader(open(f[-1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    name = r["Kernel_Name"]
    if "decode" in name or "count_selected" in name or "probe" in name:
        print("%-64s grid %8s start %10.3f ms  end %10.3f ms  dur %9.3f ms" % (name[:64], r.get("Grid_Size","?"), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
