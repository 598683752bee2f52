#!/bin/bash
# SQ instruction counters of the wavefront-mapped encoder / decoder on a batch that cannot fill the chip (512 blocks of fuzzer-style data:
# two wavefronts per CU): wave-instructions per block and per sequence, against the kernels' durations -- is a lone wavefront's time its
# instruction issue or its memory latency?     usage: bash tools/pmc_wave_small_batch.sh [blocks]
n=${1:-512}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_wave_small_batch
rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/wave_small_batch.py $n > $out/run_trace.txt 2>> $out/err.txt
i=0
for pmc in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pmc -d $out/p$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/wave_small_batch.py $n > $out/run_$i.txt 2>> $out/err.txt
done
cd $GRAFT_REPO_ROOT
python - $out $n <<'PY'
import csv, glob, json, sys, collections
out, n = sys.argv[1], int(sys.argv[2])
res = collections.defaultdict(dict)
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for kern in ('encode_fast_kernel', 'decode_kernel'):
        mine = [r for r in rows if kern in r['Kernel_Name'] and 'lane' not in r['Kernel_Name']]
        if not mine: continue
        last_id = max(int(r['Dispatch_Id']) for r in mine)          # the timed launch = the last one of that kernel
        for r in mine:
            if int(r['Dispatch_Id']) == last_id: res[kern][r['Counter_Name']] = float(r['Counter_Value'])
dur = {}
for f in glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        for kern in ('encode_fast_kernel', 'decode_kernel'):
            if kern in r['Kernel_Name'] and 'lane' not in r['Kernel_Name']: dur[kern] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
summary = {"blocks": n, "kernel_ms_last_launch": dur, "counters_last_launch": res}
json.dump(summary, open(out + '/summary.json', 'w'), indent=1)
print(json.dumps(summary, indent=1))
PY
