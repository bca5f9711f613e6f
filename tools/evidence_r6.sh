#!/bin/bash
# Round 6's evidence (run on the GPU box from the repo root): bash tools/evidence_r6.sh ; copies go to profiles/r6/
#   1. the default bench line, the C5 line, the one-process lines (1 slot, 2 slots on the one GPU)
#   2. rocprofv3 --kernel-trace --stats of the default bench command (minus the CPU leg)
#   3. PMC passes of the same command (profiles/run_profile.sh: separate --pmc runs, never combined with tracing) ->
#      profiles/summarize.py -> pmc_summary.json (bench.py's pmc_traffic / pmc_issue_model pick the newest round up by themselves)
#   4. TSVQ build: times, per-level kernel timelines, FETCH_SIZE / WRITE_SIZE of the build's kernels on uniform and N(0,1) rows
#   5. TSVQ encode metrics, host-transfer rates, the micro-benchmark behind the screen's bound
set -u
O=gpurun_out/prof_r6; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
timeout 400 python bench.py --config C5 --no-configs --no-cpu-baseline > $O/bench_c5.json 2>/dev/null
timeout 300 python bench.py --gpus 1 --one-process > $O/bench_one_process_1.json 2>/dev/null
timeout 300 python bench.py --gpus 2 --one-process --device-list 0,0 --scaling strong > $O/bench_one_process_2slots_1gpu.json 2>/dev/null
(timeout 200 python tools/tsvq_time.py; timeout 100 python tools/tsvq_time.py normal) 2>&1 | grep TSVQ > $O/tsvq_build_times.txt; cat $O/tsvq_build_times.txt
bash tools/tsvq_prof.sh tsvq_time.py c4 400 > $O/tsvq_levels_c4.txt 2>&1; tail -1 $O/tsvq_levels_c4.txt
bash tools/tsvq_prof.sh tsvq_time.py normal 400 > $O/tsvq_levels_c4_normal.txt 2>&1; tail -1 $O/tsvq_levels_c4_normal.txt
bash tools/tsvq_prof.sh tsvq_time.py normal --sum > $O/tsvq_kernels_normal_sum.txt 2>&1
bash tools/tsvq_prof.sh tsvq_time.py c4 --sum > $O/tsvq_kernels_c4_sum.txt 2>&1
timeout 200 python tools/host_xfer.py > $O/host_xfer.txt 2>&1; tail -4 $O/host_xfer.txt
timeout 200 python tools/tsvq_enc_f16.py > $O/tsvq_encode_metrics.txt 2>&1; tail -2 $O/tsvq_encode_metrics.txt
timeout 200 python tools/adc_time.py 2>&1 | grep adc > $O/adc_times.txt
timeout 300 python tools/launch_floor.py 2>&1 | grep -v amdgpu.ids > $O/launch_floor.txt
timeout 200 profiles/ubench/bin/valu_waves > $O/ubench_valu_waves.txt 2>&1
# TSVQ build traffic: FETCH_SIZE / WRITE_SIZE per kernel over one script run (3 builds each), uniform and N(0,1) rows
( cd /tmp && export TMPDIR=/tmp
  for rows in c4 normal; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/tsvq_pmc_${rows}_$c
      timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/tsvq_pmc_${rows}_$c -o pmc -- python3 $R/tools/tsvq_time.py $rows > /dev/null 2>&1 || echo "tsvq pmc $rows $c failed" >> $R/$O/errors.txt
    done
  done
  python3 - "$R/$O" <<'PY'
import collections, csv, glob, json, re, sys
out = sys.argv[1]
res = {}
for rows in ("c4", "normal"):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); builds = 0
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"/tmp/tsvq_pmc_{rows}_{c}/**/*counter_collection.csv", recursive=True):
            nb = 0
            for r in csv.DictReader(open(f)):
                m = re.search(r"(k_[A-Za-z_0-9]+)", r["Kernel_Name"])
                if not m: continue
                if m.group(1) == "k_build_init": nb += 1
                per[m.group(1)][c] += float(r["Counter_Value"])
            builds = max(builds, nb)
    builds = max(builds, 1)
    tot_f = sum(v["FETCH_SIZE"] for v in per.values()) / builds; tot_w = sum(v["WRITE_SIZE"] for v in per.values()) / builds
    # KiB units; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md)
    res[rows] = {"builds": builds, "fetch_kib_per_build_raw": tot_f, "write_kib_per_build": tot_w,
                 "hbm_bytes_per_build": (2.0 * tot_f + tot_w) * 1024.0, "algorithmic_bytes": 4.0 * 1e6 * 128 * 17,
                 "kernels": {k: {c: v[c] / builds for c in v} for k, v in sorted(per.items(), key=lambda kv: -kv[1]["FETCH_SIZE"])[:14]}}
    res[rows]["traffic_over_algorithmic"] = res[rows]["hbm_bytes_per_build"] / res[rows]["algorithmic_bytes"]
json.dump(res, open(out + "/tsvq_build_pmc_summary.json", "w"), indent=1)
print({k: (round(v["hbm_bytes_per_build"] / 1e9, 3), round(v["traffic_over_algorithmic"], 3)) for k, v in res.items()})
PY
)
# kernel trace + PMC passes of the default bench command
bash profiles/run_profile.sh r6 > $O/run_profile.log 2>&1
python3 profiles/summarize.py gpurun_out/prof_r6 $O/final > $O/summarize.log 2>&1 || true
# the raw per-dispatch csv files are tens of MB each: only the summaries travel back (gpurun merges <= 64 MiB)
cp $O/trace/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/pmc_*/ $O/trace
du -sh $O; ls $O | head -60
