#!/bin/bash
# kernel-trace + stats of an arbitrary python entry:  bash profiles/run_trace.sh <tag> bench_configs.py C4
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_$TAG
mkdir -p "$OUT"
SCRIPT=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 "$SCRIPT" "$@" > "$OUT/stdout.txt" 2> "$OUT/stderr.txt"
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
head -25 "$f" | cut -c1-200
