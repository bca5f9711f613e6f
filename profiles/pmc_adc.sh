#!/bin/bash
# PMC passes over the ADC search (tools/adc_time.py 64): bash profiles/pmc_adc.sh <out dir under gpurun_out>
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-pmc_adc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
           "GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -o pmc -- python3 $REPO/tools/adc_time.py 64 > "$OUT/g$i.txt" 2> "$OUT/g$i.err" || echo "group $i failed" >> "$OUT/errors.txt"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_adc_\w+)(<[^>]*>)?", r["Kernel_Name"])
        if not m: continue
        a = acc[m.group(1) + (m.group(2) or "").replace(" ", "")][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
res = {k: {c: {"avg_per_launch": s / max(1, n), "launches": n} for c, (s, n) in sorted(cs.items())} for k, cs in acc.items()}
json.dump(res, open(out + "/adc_pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, cs in res.items():
    print(k)
    for c, v in cs.items(): print("   %-28s %.5g" % (c, v["avg_per_launch"]))
PY
rm -rf "$OUT"/g*/
