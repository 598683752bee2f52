#!/bin/bash
# Runs on the GPU box: the -m gpu suite under rocprofv3 --kernel-trace --stats, so that the summary lists which
# decode/encode kernel instantiations the parity tests actually launched.   -> gpurun_out/<tag>/
tag=${1:-r02_tests}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $out/trace -o tests --output-format csv -- \
    python -m pytest $GRAFT_REPO_ROOT/tests -x -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
cd $GRAFT_REPO_ROOT
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/trace
tail -5 $out/pytest.log
python - "$out" <<'PY'
import csv, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/kernel_stats.csv")))
with open(out + "/kernel_stats_summary.txt", "w") as f:
    f.write("rocprofv3 --kernel-trace --stats -- python -m pytest tests -m gpu\n")
    f.write("%-90s %8s %14s %12s\n" % ("kernel", "calls", "total_ns", "avg_ns"))
    for r in rows:
        f.write("%-90s %8s %14s %12.0f\n" % (r["Name"][:90], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"])))
print(open(out + "/kernel_stats_summary.txt").read())
PY
