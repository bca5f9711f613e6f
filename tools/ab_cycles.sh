#!/bin/bash
# time AND core cycles of the screen kernel for two library builds on one box: is a change that removes stalls paid back
# in clock (power management) instead of time?   bash tools/ab_cycles.sh ab/libvqhip_head.so
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for L in "$1" ""; do
  if [ -n "$L" ]; then export VQHIP_LIB_PATH=$REPO/$L; else unset VQHIP_LIB_PATH; fi
  rm -rf /tmp/abc_t /tmp/abc_p
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abc_t -o t -- python3 $REPO/tools/ab_screen.py c2 > /tmp/abc.log 2>&1
  timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/abc_p -o p -- python3 $REPO/tools/ab_screen.py c2 >> /tmp/abc.log 2>&1
  echo "== ${L:-new}"
  grep "x32p<16, 8, false>" $(find /tmp/abc_t -name "*kernel_stats.csv") | awk -F'",' '{print "calls,total_ns,avg_ns,...: " $2}'
  python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('/tmp/abc_p/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'x32p<16, 8, false>' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    v=sorted(v); print('  %-20s n=%d median %.0f'%(k,len(v),v[len(v)//2]))
PY
done
