#!/bin/bash
# per-kernel timeline of the last Lloyd iterations of tools/km_time.py's device-driven run (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_km
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_km -o t -- python3 $GRAFT_REPO_ROOT/tools/km_time.py > /tmp/o_km.txt 2>&1
f=$(find /tmp/prof_km -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, re, sys
tr = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in tr:
    m = re.search(r'(k_\w+)(<[^>]*>)?', r['Kernel_Name'])
    if m: ks.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), m.group(0)))
ks.sort()
sel = ks[-40:]
t0 = sel[0][0]
for s, e, n in sel:
    print(f"{(s - t0) / 1e3:9.1f} us  {n:50s} {(e - s) / 1e3:8.1f} us")
PY
tail -2 /tmp/o_km.txt
