#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the decode kernels for the default bench workload (separate passes).
tag=$1; dist=$2; blocks=$3; shift 3
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  env "$@" timeout 300 rocprofv3 --pmc $pmc -d $out/p$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-extras --no-cpu --blocks $blocks --dist $dist > /dev/null 2>> $out/err.txt
done
cd $GRAFT_REPO_ROOT
python - $out $blocks <<'PY'
import csv,glob,collections,json,sys
out=sys.argv[1]; res=collections.defaultdict(float)
for f in glob.glob(out+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'decode' in r['Kernel_Name']: res[r['Counter_Name']]+=float(r['Counter_Value'])
d=dict(res); d['launches_summed']=2; d['blocks_per_launch']=int(sys.argv[2])
json.dump(d, open(out+'/summary.json','w'), indent=1)
print(out, json.dumps(d))
PY
