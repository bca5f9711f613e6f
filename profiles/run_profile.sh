#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline block refers to.  Run on the GPU box
# from the repo root:  bash profiles/run_profile.sh <tag>
# Kernel trace and each PMC group are separate passes (never combined with tracing); every pass under `timeout` (a
# counter group that aborts under the tool has hung until the box's own limit: 20 GPU-minutes for nothing).
set -u
TAG=${1:-r1}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $grp | tr ' ' '+' | cut -c1-40)
  timeout 900 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o pmc -- $BENCH > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || echo "pmc group '$grp' failed" >> "$OUT/errors.txt"
done
timeout 120 rocprofv3 -L > "$OUT/counters_list.txt" 2>&1 || true
find "$OUT" -name "*.csv" | head -50 > "$OUT/files.txt"
du -sh "$OUT" >> "$OUT/files.txt"
