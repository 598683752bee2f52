#!/bin/bash
# PMC passes for the decode kernels of the default bench workload (separate runs, counters only).
# Usage: [KERNEL_FILTER=encode_fast_lane] [PMC_QUICK=1] bash tools/pmc_decoder.sh <tag> <dist> <blocks> [ENV=VAL...]     (PMC_QUICK: the two SQ passes only)
tag=$1; dist=$2; blocks=$3; shift 3
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
if [ -n "$PMC_QUICK" ]; then
  passes=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE")
else
  passes=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE")
fi
for pmc in "${passes[@]}"; do
  i=$((i+1))
  env "$@" timeout 300 rocprofv3 --pmc $pmc -d $out/p$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 ${BENCH_ARGS:---no-extras} --no-cpu --blocks $blocks --dist $dist > /dev/null 2>> $out/err.txt
done
cd $GRAFT_REPO_ROOT
python - $out $blocks <<'PY'
import csv,glob,collections,json,sys,os
out=sys.argv[1]; res=collections.defaultdict(float); launches=collections.Counter()
for f in glob.glob(out+'/p*/**/*counter_collection.csv', recursive=True):
    seen=set()
    for r in csv.DictReader(open(f)):
        if os.environ.get('KERNEL_FILTER', 'decode') in r['Kernel_Name']:
            res[r['Counter_Name']]+=float(r['Counter_Value']); seen.add(r['Dispatch_Id'])
    for c in set(r2 for r2 in res): pass
res['_note']=0
d=dict(res); d.pop('_note')
d['launches_summed']=2; d['blocks_per_launch']=int(sys.argv[2])
json.dump(d, open(out+'/summary.json','w'), indent=1)
print(out, json.dumps(d))
PY
