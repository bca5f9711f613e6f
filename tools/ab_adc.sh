#!/bin/bash
# ADC A/B on one box: the full pass (VQHIP_ADC_FAST=0) against the one-scan schedule (a row per lane: VQHIP_ADC_LQ1=1; default: two
# lanes per row), then per-kernel times of the default
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
echo "== full pass"; VQHIP_ADC_FAST=0 python3 $REPO/tools/adc_time.py 2>&1 | tail -3
echo "== one scan, 8 queries per batch, a row per lane"; VQHIP_ADC_LQ1=1 python3 $REPO/tools/adc_time.py 2>&1 | tail -3
echo "== one scan (default)"; python3 $REPO/tools/adc_time.py 2>&1 | tail -3
for NQ in 64 8 1; do
rm -rf /tmp/adc_t
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/adc_t -o t -- python3 $REPO/tools/adc_time.py $NQ > /tmp/adc.log 2>&1
echo "== kernels, $NQ queries"
python3 - <<'PY'
import csv,glob,re
f=glob.glob('/tmp/adc_t/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'adc' in r['Name']:
        nm=re.search(r'(k_adc_\w+(<[^>]*>)?)', r['Name']).group(1)
        print('  %-32s calls %5s avg %9.1f us  min %9.1f max %9.1f' % (nm, r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
