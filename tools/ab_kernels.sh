#!/bin/bash
# per-kernel average times of tools/ab_screen.py <which> for several library builds on one box ("new" = the tree's own)
#   bash tools/ab_kernels.sh c3 ab/libvqhip_base.so ab/libvqhip_x.so new
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
WHICH=$1; shift
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  if [ "$L" != "new" ]; then export VQHIP_LIB_PATH=$REPO/$L; else unset VQHIP_LIB_PATH; fi
  rm -rf /tmp/abk_t
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk_t -o t -- python3 $REPO/tools/ab_screen.py $WHICH > /tmp/abk.log 2>&1
  echo "== $L"
  python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/abk_t/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    if 'x32p' in r['Name'] or 'recheck' in r['Name'] or 'accumulate' in r['Name'] or 'reduce' in r['Name']:
        print('  %-74s calls %5s avg %10.1f us' % (r['Name'][39:113], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
