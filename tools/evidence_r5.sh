#!/bin/bash
# Round 5's evidence (run on the GPU box from the repo root): bash tools/evidence_r5.sh <tag>; copies go to profiles/r5/
set -u
TAG=${1:-a}
O=gpurun_out/prof_r5$TAG; mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
timeout 400 python bench.py --config C5 --no-configs --no-cpu-baseline > $O/bench_c5.json 2>/dev/null
timeout 300 python bench.py --gpus 1 --one-process > $O/bench_one_process_1.json 2>/dev/null
timeout 300 python bench.py --gpus 2 --one-process --device-list 0,0 --scaling strong > $O/bench_one_process_2slots_1gpu.json 2>/dev/null
(timeout 200 python tools/tsvq_time.py; timeout 100 python tools/tsvq_time.py normal; timeout 100 env VQHIP_TSVQ_CHAIN=1 python tools/tsvq_time.py normal) 2>&1 | grep TSVQ > $O/tsvq_build_times.txt; cat $O/tsvq_build_times.txt
(timeout 100 env VQHIP_TSVQ_DEBUG=1 python tools/tsvq_time.py normal 2>&1 | grep -A1 "level . mean" | head -20) > $O/tsvq_table_stats_normal.txt
bash tools/tsvq_prof.sh tsvq_time.py c4 400 > $O/tsvq_levels_c4.txt 2>&1; tail -2 $O/tsvq_levels_c4.txt
bash tools/tsvq_prof.sh tsvq_time.py normal 400 > $O/tsvq_levels_c4_normal.txt 2>&1; tail -2 $O/tsvq_levels_c4_normal.txt
bash tools/tsvq_prof.sh tsvq_time.py normal --sum > $O/tsvq_kernels_normal_sum.txt 2>&1
bash tools/tsvq_prof.sh tsvq_time.py c4 --sum > $O/tsvq_kernels_c4_sum.txt 2>&1
timeout 200 python tools/host_xfer.py > $O/host_xfer.txt 2>&1; tail -4 $O/host_xfer.txt
timeout 200 python tools/tsvq_enc_f16.py > $O/tsvq_encode_metrics.txt 2>&1; tail -2 $O/tsvq_encode_metrics.txt
# rocprofv3 --kernel-trace --stats of the default bench command (minus the CPU leg)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o trace -- python3 $R/bench.py --no-cpu-baseline > $R/$O/bench_under_trace.json 2> /tmp/trace.err
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $R/$O/kernel_stats.csv
head -12 $R/$O/kernel_stats.csv
