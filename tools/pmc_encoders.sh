#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the lane encoders, per launch and per block.
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_encoders
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pmc -d $out/f$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/enc_rate.py 262144 > /dev/null 2>> $out/err.txt
  timeout 300 rocprofv3 --pmc $pmc -d $out/h$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/hc_rate.py 131072 16 > /dev/null 2>> $out/err.txt
done
cd $GRAFT_REPO_ROOT
python - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(list)
for f in glob.glob(out + '/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'encode_fast_lane_kernel' in k or 'encode_hc_lane_kernel' in k:
            rows[(k.split('(')[0], r['Counter_Name'])].append((int(r['Grid_Size']), float(r['Counter_Value'])))
res = {}
for (k, c), v in rows.items():
    big = max(g for g, _ in v)
    vals = [x for g, x in v if g == big]          # the full-size launches only
    res.setdefault(k, {})[c] = {"launches": len(vals), "KiB_per_launch_mean": sum(vals) / len(vals), "grid": big}
json.dump(res, open(out + '/summary.json', 'w'), indent=1)
print(json.dumps(res))
PY
