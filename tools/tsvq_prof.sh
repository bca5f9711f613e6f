#!/bin/bash
# usage: tsvq_prof.sh <script> <arg> [lines | --sum]: per-kernel timeline (or totals per kernel) of the last build
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o t -- python3 $GRAFT_REPO_ROOT/tools/$1 $2 > /tmp/o.txt 2>&1
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
if [ "$3" == "--sum" ]; then python3 $GRAFT_REPO_ROOT/tools/tsvq_levels.py $f --sum; else python3 $GRAFT_REPO_ROOT/tools/tsvq_levels.py $f fs_ 2>&1 | head -${3:-20}; fi
