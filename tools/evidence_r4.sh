set -u
O=gpurun_out/prof_r4c; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_piped.py -x -q 2>&1 | tail -2
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json | head -c 300; echo
timeout 400 python bench.py --config C5 --no-configs --no-cpu-baseline > $O/bench_c5.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/bench_c5.json').read().strip().splitlines()[-1]);print('C5',d['value'],d['ms_per_step'],d.get('kmeans_ms_per_iter'))"
(timeout 200 python tools/tsvq_time.py; timeout 100 python tools/tsvq_time.py normal) 2>&1 | grep TSVQ > $O/tsvq_build_times.txt; cat $O/tsvq_build_times.txt
timeout 200 python tools/host_xfer.py > $O/host_xfer.txt 2>&1; tail -8 $O/host_xfer.txt
timeout 200 python tools/tsvq_enc_f16.py > $O/tsvq_encode_metrics.txt 2>&1; tail -4 $O/tsvq_encode_metrics.txt
bash tools/c1_prof.sh > $O/c1_trace.txt 2>&1; tail -12 $O/c1_trace.txt
bash tools/tsvq_prof.sh tsvq_time.py c4 400 > $O/tsvq_levels_c4.txt 2>&1; tail -2 $O/tsvq_levels_c4.txt
