#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / TCC hit-miss of the decode kernels only, for one distribution (separate rocprofv3 --pmc passes).
# Usage: bash tools/pmc_decode_traffic.sh <dist> [blocks]
dist=${1:-3}; blocks=${2:-1048576}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_decode_traffic_D$dist
rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pmc -d $out/p$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-extras --no-cpu --blocks $blocks --dist $dist > $out/bench_$i.json 2>> $out/err.txt
done
cd $GRAFT_REPO_ROOT
python - $out $dist $blocks <<'PY'
import csv, glob, json, sys, collections
out, dist, blocks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
vals = collections.defaultdict(list)
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    lane = lambda r: 'decode_lane4_kernel' in r['Kernel_Name'] or 'decode_lane3_kernel' in r['Kernel_Name'] or 'decode_lane_kernel' in r['Kernel_Name']
    rows = [r for r in csv.DictReader(open(f)) if lane(r) or 'decode_kernel' in r['Kernel_Name']]
    big = max(int(r['Grid_Size']) for r in rows if lane(r))
    per = collections.defaultdict(float); n = collections.Counter()
    for r in rows:
        if lane(r) and int(r['Grid_Size']) == big:
            per[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    for c, v in per.items():
        vals[c] = v / n[c]
b = json.load(open(out + '/bench_1.json'))
res = {"dist": dist, "blocks": blocks, "per_launch": dict(vals), "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"],
       "mean_kernel_ms": b["roofline"]["mean_kernel_ms"], "note": "FETCH_SIZE / WRITE_SIZE in KiB; lane decoder's full-size launches only"}
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    res["traffic_bytes"] = int((vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
    res["traffic_over_algorithmic"] = round(res["traffic_bytes"] / res["algorithmic_bytes_per_launch"], 3)
    res["traffic_GBps"] = round(res["traffic_bytes"] / (res["mean_kernel_ms"] / 1e3) / 1e9, 1)
json.dump(res, open(out + '/summary.json', 'w'), indent=1)
print(json.dumps(res))
PY
