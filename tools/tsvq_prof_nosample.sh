#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
export VQHIP_TSVQ_SAMPLE=1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o t -- python3 $GRAFT_REPO_ROOT/tools/tsvq_time.py c4 > /tmp/o.txt 2>&1
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/tsvq_levels.py $f --sum
