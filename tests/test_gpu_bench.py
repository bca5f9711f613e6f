"""bench.py end to end on the GPU box: the self-launch path and the N > 1 plumbing with two ranks
sharing the one GPU (RCCL refuses two ranks on one device, so the collective there is gloo with the slab staged on
the host: everything else -- sharding, global init rows, barriers, max-over-ranks timing, the relay of rank 0's
line -- is the code an 8-GPU run executes)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(argv, env=None, timeout=900):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                       timeout=timeout, env=e)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


SMALL = ["--steps", "2", "--warmup", "1", "--kmeans-iters", "3", "--no-cpu-baseline", "--no-configs"]


def test_one_rank_through_the_spawn_path():
    line = _bench(["--gpus", "1", "--rows", "200000"] + SMALL, env={"VQ_BENCH_SPAWN": "1"})
    assert line["n_gpus"] == 1 and line["rccl_world"] == 1 and line["scaling"] == "weak"
    assert line["kmeans_valid"] and line["kmeans_iters_timed"] == [3, 3] and line["kmeans_active_subspaces"] == 8.0
    assert line["kmeans_counts_sum_per_subspace"] == [200000, 200000]
    assert line["value"] > 1e7 and 0 < line["roofline"]["frac"] < 1.2
    direct = _bench(["--gpus", "1", "--rows", "200000"] + SMALL)
    assert direct["codes_checksum_rank0"] == line["codes_checksum_rank0"]
    assert direct["codebooks_abs_sum"] == line["codebooks_abs_sum"]


def test_two_ranks_sharing_the_gpu_weak_and_strong():
    env = {"VQ_BENCH_SHARE_GPU": "1"}
    one = _bench(["--gpus", "1", "--rows", "200001"] + SMALL)
    strong = _bench(["--gpus", "2", "--collective", "gloo", "--scaling", "strong", "--rows", "200001"] + SMALL, env=env)
    assert strong["n_gpus"] == 2 and strong["scaling"] == "strong"
    assert strong["config"]["rows_global"] == 200001 and strong["config"]["rows_per_gpu"] == 100001  # rank 0 of an uneven split
    assert strong["kmeans_counts_sum_per_subspace"] == [200001, 200001]  # the all-reduce summed both shards
    # same global job, two shards: the codebooks agree up to the summation order of the f64 slabs
    assert abs(strong["codebooks_abs_sum"] - one["codebooks_abs_sum"]) <= 1e-6 * one["codebooks_abs_sum"]
    weak = _bench(["--gpus", "2", "--collective", "gloo", "--rows", "100000"] + SMALL, env=env)
    assert weak["scaling"] == "weak" and weak["config"]["rows_global"] == 200000
    assert weak["kmeans_counts_sum_per_subspace"] == [200000, 200000]
    assert weak["value"] > 0 and weak["kmeans_valid"]


def test_two_ranks_strong_and_c5_blocks():
    """what an N > 1 line carries beside the headline: the same job strong-scaled and BASELINE configs[4]'s per-GPU share
    (12.5M x 128 rows per rank, m = 16) -- here two ranks on the one GPU, the all-reduce over gloo"""
    line = _bench(["--gpus", "2", "--collective", "gloo", "--rows", "100000", "--steps", "2", "--warmup", "1", "--kmeans-iters", "3",
                   "--no-cpu-baseline"], env={"VQ_BENCH_SHARE_GPU": "1"}, timeout=1200)
    s, c5 = line["strong_C2"], line["weak_C5"]
    assert s["scaling"] == "strong" and s["rows_global"] == 100000 and s["rows_this_rank"] == 50000
    assert s["kmeans_counts_sum_per_subspace"] == [100000, 100000] and s["kmeans_valid"]
    assert c5["rows_global"] == 25_000_000 and c5["m"] == 16 and c5["kmeans_counts_sum_per_subspace"] == [25_000_000] * 2
    assert c5["kmeans_valid"] and c5["encode_vectors_per_s"] > 1e8
    assert "configs" not in line  # the C1 / C3 / C4 block belongs to the one-GPU line


def test_one_process_launcher_two_slots_on_the_gpu():
    """`bench.py --gpus 2 --one-process`: one process, the library's own ranks (worker threads), here both on device 0; the
    same global job as one rank -- counts prove the exchange summed both blocks, the codes of the first block's rows and
    the codebooks agree with the one-rank line"""
    one = _bench(["--gpus", "1", "--rows", "200001", "--scaling", "strong"] + SMALL)
    two = _bench(["--gpus", "2", "--one-process", "--device-list", "0,0", "--scaling", "strong", "--rows", "200001"] + SMALL)
    assert two["n_gpus"] == 2 and two["comm_world"] == 2 and two["devices"] == [0, 0]
    assert "in-process" in two["kmeans_collective"] and two["launcher"].startswith("one process")
    assert two["config"]["rows_global"] == 200001 and two["config"]["rows_per_gpu"] == 100001
    assert two["kmeans_counts_sum_per_subspace"] == [200001, 200001] and two["config"]["kmeans_valid"]
    assert abs(two["codebooks_abs_sum"] - one["codebooks_abs_sum"]) <= 1e-6 * one["codebooks_abs_sum"]
    assert two["value"] > 1e7 and two["config"]["kmeans_iter_per_s"] > 0
    solo = _bench(["--gpus", "1", "--one-process", "--rows", "200001", "--scaling", "strong"] + SMALL)
    assert solo["comm_world"] == 1 and solo["codebooks_abs_sum"] == one["codebooks_abs_sum"]
