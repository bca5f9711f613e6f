#!/bin/bash
# Issue / wait counters of one kernel family (default: the X32 screens) over a short bench.py run.  One PMC group per pass.
#   bash profiles/pmc_kernel.sh <tag> [extra bench.py args]
set -u
TAG=${1:-x}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmck_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --no-cpu-baseline --no-configs ${*:---steps 20 --warmup 2}"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -o pmc -- $BENCH > "$OUT/g$i.json" 2> "$OUT/g$i.err" || echo "group $i failed" >> "$OUT/errors.txt"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[A-Za-z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
        if not m or "screen_bf16" not in m.group(1): continue
        k = m.group(1) + (m.group(2) or "").replace(" ", "")
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
res = {}
for k, cs in acc.items():
    # rows are per (dispatch, dimension instance): report the per-dispatch total = sum / dispatches
    res[k] = {}
    for c, (s, nrows) in sorted(cs.items()):
        res[k][c] = s / max(1, nrows)   # average per launch
json.dump(res, open(out + "/summary_avg_per_launch.json", "w"), indent=1)
for k, cs in res.items():
    print(k)
    for c, v in cs.items(): print("   %-32s %.4g" % (c, v))
PY
