#!/bin/bash
# k_gather_f16 (encode's f16 rows) and k_decode_f32 with the codebooks gathered from L2 (VQHIP_DECODE_LDS=0) or held in LDS: kernel times
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  export VQHIP_DECODE_LDS=$V
  rm -rf /tmp/gt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gt -o t -- python3 $REPO/tools/decode_time.py > /tmp/gt.log 2>&1
  echo "== VQHIP_DECODE_LDS=$V"
  python3 - <<'PY'
import csv,glob,re
f=glob.glob('/tmp/gt/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'gather_f16' in r['Name'] or 'decode_f32' in r['Name'] or 'dequant' in r['Name']:
        nm=re.search(r'(k_\w+(<[^>]*>)?)', r['Name']).group(1)
        print('  %-28s calls %5s avg %9.1f us  min %9.1f max %9.1f' % (nm, r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
