#!/bin/bash
# Runs on the GPU box (via gpurun): per-kernel timing of the bench's headline workload under rocprofv3.
# Usage: bash tools/profile_bench.sh <round-tag>   -> gpurun_out/<tag>/...
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace -o bench --output-format csv -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-extras --no-cpu > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
cd $GRAFT_REPO_ROOT
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
python - "$out" <<'PY'
import csv, sys, collections, glob
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/kernel_stats.csv")))
with open(out + "/kernel_stats_summary.txt", "w") as f:
    f.write("rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-extras --no-cpu\n")
    f.write("(headline workload only, so that the per-kernel average IS the timed decode launch; the default bench.py adds the other distributions, the encoders and the CPU baseline after the timed region)\n")
    f.write("%-70s %8s %14s %14s %8s\n" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
    for r in rows:
        f.write("%-70s %8s %14s %14.0f %8s\n" % (r["Name"][:70], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
print(open(out + "/kernel_stats_summary.txt").read())
# per-dispatch durations of the decode kernels in launch order (to read off the timed steps)
for t in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    with open(out + "/decode_dispatches.txt", "w") as f:
        for r in csv.DictReader(open(t)):
            if "decode" in r["Kernel_Name"]:
                f.write("%s grid=%s %0.3f ms\n" % (r["Kernel_Name"][:60], r.get("Grid_Size", "?"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
tail -c 1500 $out/bench_under_rocprof.json
