#!/bin/bash
# SQ counters of the LZ4HC kernels (chain builder, length fill, lane kernel): separate rocprofv3 --pmc passes over tools/hc_gen_ab.py.
# Usage: bash tools/pmc_hc.sh <tag> [blocks] [cfg] [dist]   -> gpurun_out/pmc_hc_<tag>/summary.json
tag=${1:-hc}; blocks=${2:-65536}; cfg=${3:-4:16}; dist=${4:-2}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_hc_$tag
rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for pmc in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pmc -d $out/p$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/hc_gen_ab.py $blocks "$cfg" "$dist" > $out/run_$i.txt 2>&1
done
cd $GRAFT_REPO_ROOT
python - $out $blocks <<'PY'
import csv, glob, json, sys, collections
out, blocks = sys.argv[1], int(sys.argv[2])
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        name = r['Kernel_Name']
        for key in ('hc_nat_chain_kernel', 'hc_lcp_fill_kernel', 'encode_hc_lcp_kernel', 'encode_hc_nat_kernel'):
            if key in name:
                res[key][(r['Counter_Name'], int(r['Grid_Size']))].append(float(r['Counter_Value']))
summary = {}
for key, d in res.items():
    big = max(g for (_, g) in d)
    summary[key] = {c: sum(v) / len(v) for (c, g), v in d.items() if g == big}
    summary[key]['grid'] = big
json.dump({"blocks": blocks, "per_full_size_launch": summary}, open(out + '/summary.json', 'w'), indent=1)
print(json.dumps(summary, indent=1))
PY
