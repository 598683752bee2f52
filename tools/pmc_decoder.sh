#!/bin/bash
# PMC passes for one decoder mapping.  Usage: bash tools/pmc_decoder.sh <tag> <dist> ENV=VAL...
tag=$1; dist=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for pmc in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  env "$@" timeout 300 rocprofv3 --pmc $pmc -d $out/p$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-extras --no-cpu --blocks 524288 --dist $dist > /dev/null 2>> $out/err.txt
done
cd $GRAFT_REPO_ROOT
python - $out <<'PY'
import csv,glob,collections,json,sys
out=sys.argv[1]; res=collections.defaultdict(float)
for f in glob.glob(out+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'decode' in r['Kernel_Name']: res[r['Counter_Name']]+=float(r['Counter_Value'])
json.dump(res, open(out+'/summary.json','w'), indent=1)
print(out, json.dumps(res))
PY
