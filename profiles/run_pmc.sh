#!/bin/bash
# PMC passes (each its own run, no tracing) over any python entry point:
#   bash profiles/run_pmc.sh <tag> <kernel-name-substring> <script.py> [args...]
TAG=$1; FILTER=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
SCRIPT=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -o pmc -- python3 "$SCRIPT" "$@" > "$OUT/g$i.out" 2> "$OUT/g$i.err" || echo "group $i ($grp) failed" >> "$OUT/errors.txt"
done
python3 - "$OUT" "$FILTER" <<'PY'
import csv,glob,re,collections,sys,json
out,flt=sys.argv[1],sys.argv[2]
summary=collections.defaultdict(dict)
for f in sorted(glob.glob(out+'/g*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        m=re.search(r'(k_[A-Za-z_0-9]+)',r['Kernel_Name'])
        if m and (flt in m.group(1)): agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        for c,x in v.items(): summary[k][c]={"avg_per_launch":sum(x)/len(x),"launches":len(x)}
json.dump(summary,open(out+'/pmc_summary.json','w'),indent=1)
for k,v in summary.items():
    print(k)
    for c,x in v.items(): print(f"   {c:32s} {x['avg_per_launch']:16.1f}  ({x['launches']} launches)")
if glob.glob(out+'/errors.txt'): print(open(out+'/errors.txt').read())
PY
