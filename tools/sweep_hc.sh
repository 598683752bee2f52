mkdir -p gpurun_out
python - <<'PY' 2>>gpurun_out/sweep6.err | tee -a gpurun_out/sweep6.txt
import os, sys, time
sys.path.insert(0, '.')
import torch
from lz4net_amd import batch
torch.cuda.set_device(0)
def ev(fn):
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); b.synchronize(); return a.elapsed_time(b)
for dist in (2,3):
    n=1<<16
    raw=batch.synth(dist, 5, 0, n); comp=torch.empty((n,batch.BOUND_STRIDE),dtype=torch.uint8,device='cuda')
    ref=None
    for name,env in (("wave",{"LZ4HIP_HC":"wave"}),("lane4",{"LZ4HIP_HC":"lane","LZ4HIP_HC_WAVES_PER_CU":"4"}),("lane8",{"LZ4HIP_HC":"lane","LZ4HIP_HC_WAVES_PER_CU":"8"}),("lane2",{"LZ4HIP_HC":"lane","LZ4HIP_HC_WAVES_PER_CU":"2"})):
        os.environ.update(env)
        m = n if name!="wave" else 1<<14
        batch.encode(raw[:256], batch.BLOCK, comp[:256], batch.BOUND, hc=True); torch.cuda.synchronize()
        h={}
        ms=ev(lambda: h.setdefault('c', batch.encode(raw[:m], batch.BLOCK, comp[:m], batch.BOUND, hc=True)))
        clen=h['c']; s=int(clen.to(torch.int64).sum().item())
        back=torch.empty((m,65536),dtype=torch.uint8,device='cuda'); used=batch.decode(comp[:m], clen, back, batch.BLOCK)
        ok = bool((used==clen).all()) and batch.count_mismatches(raw[:m], back, batch.BLOCK)==0
        print(f"dist={dist} hc={name}: {m*65536/ms/1e6:.3f} GB/s  ratio {s/(m*65536):.4f} ok={ok}", flush=True)
PY
