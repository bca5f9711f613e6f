#!/bin/bash
# quick PMC passes over a short encode-only bench run: bash profiles/run_pmc_quick.sh <tag> [env assignments]
TAG=${1:-q}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
BENCH="python3 $REPO/bench.py --steps 4 --warmup 1 --kmeans-iters 1 --no-cpu-baseline"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -o pmc -- $BENCH > "$OUT/g$i.json" 2> "$OUT/g$i.err" || echo "group $i failed" >> "$OUT/errors.txt"
done
python3 - "$OUT" <<'PY'
import csv,glob,re,collections,sys
out=sys.argv[1]
for f in sorted(glob.glob(out+'/g*/pmc_counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        m=re.search(r'(k_[a-z_0-9]+)',r['Kernel_Name'])
        if m and ('screen' in m.group(1)): agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(k, {c:round(sum(x)/len(x)/1e6,2) for c,x in v.items()}, '(millions per launch)')
PY
