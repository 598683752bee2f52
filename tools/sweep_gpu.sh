mkdir -p gpurun_out/prof
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/counters.txt 2>&1
for kb in 0 5 10 20 40 80 160; do
  echo "== LDS_KB=$kb" >> gpurun_out/sweep.txt
  LZ4HIP_LANE_LDS_KB=$kb timeout 200 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu --decoder lane --blocks 524288 2>>gpurun_out/sweep.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['mean_kernel_ms'])" >> gpurun_out/sweep.txt
done
for d in 3; do for kb in 0 20 40; do
  echo "== dist=$d LDS_KB=$kb" >> gpurun_out/sweep.txt
  LZ4HIP_LANE_LDS_KB=$kb timeout 200 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu --decoder lane --blocks 524288 --dist $d 2>>gpurun_out/sweep.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['mean_kernel_ms'])" >> gpurun_out/sweep.txt
done; done
cat gpurun_out/sweep.txt
# PMC passes (separate runs, counters only)
cd /tmp
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/$tag -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-extras --no-cpu --decoder lane --blocks 524288 > /dev/null 2>> $GRAFT_REPO_ROOT/gpurun_out/prof/err.txt
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/prof/*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]; agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    print(f)
    for k,v in agg.items():
        if 'decode' in k or 'encode' in k: print('  ',k, dict(v))
PY
