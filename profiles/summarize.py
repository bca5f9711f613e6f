#!/usr/bin/env python3
"""Condenses a profiles/run_profile.sh output directory into the files kept under profiles/rN/:
    python3 profiles/summarize.py gpurun_out/prof_<tag> profiles/r1/<prefix>
writes <prefix>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, as is),
<prefix>_bench_under_trace.json and <prefix>_pmc_summary.json ({kernel: {counter: avg per launch}})."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

src, prefix = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(prefix), exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], prefix + "_kernel_stats.csv")
if os.path.exists(os.path.join(src, "trace_bench.json")):
    shutil.copy(os.path.join(src, "trace_bench.json"), prefix + "_bench_under_trace.json")
summary = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*counter_collection.csv"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[A-Za-z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
        if m:
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if m.group(2):  # per instantiation too: "k_name<16,8,1,0,false>" (spaces and the u suffix of unsigned arguments dropped)
                full = m.group(1) + re.sub(r"(\d)u\b", r"\1", m.group(2).replace(" ", ""))
                agg[full][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        for c, x in v.items():
            # one instantiation serves launches of very different sizes (the 32 MB chunks of the host paths next to the
            # 1M-row launch the bench line is about): `avg_largest` averages the launches within 20 % of the largest value
            big = [y for y in x if y >= 0.8 * max(x)] if max(x) > 0 else x
            summary[k][c] = {"avg_per_launch": sum(x) / len(x), "launches": len(x), "avg_largest": sum(big) / len(big), "launches_largest": len(big)}
json.dump(summary, open(prefix + "_pmc_summary.json", "w"), indent=1, sort_keys=True)
print("kernels with counters:", sorted(summary))
