cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o t -- python3 $GRAFT_REPO_ROOT/tools/c1_trace.py > /tmp/o.txt 2>&1
cat /tmp/o.txt | tail -3
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),(re.search(r'(k_\w+(<[^>]*>)?)',r['Kernel_Name']) or re.search(r'(\w+)',r['Kernel_Name'])).group(1)[:70]) for r in rows]
ks.sort()
# print the last 40 kernels with start offsets
t0=ks[-60][0]
for s,e,n in ks[-60:]:
    print(f"{(s-t0)/1e3:9.1f} us {(e-s)/1e3:7.1f} us  {n}")
PY
