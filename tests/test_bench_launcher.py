"""bench.py as its own launcher: `python bench.py --gpus N` without torchrun starts N fresh rank
processes before anything touches a GPU.  CPU tests of the plumbing with a stub worker; the real thing runs in
tests/test_gpu_bench.py."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub(tmp_path, body):
    p = tmp_path / "stub_worker.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def _run_launcher(tmp_path, n, argv, body, env=None):
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        rc = bench.launch_ranks({n}, {argv!r}, worker={_stub(tmp_path, body)!r}, grace_s=1.0)
        assert 'torch' not in sys.modules and 'vq_amd' not in sys.modules, 'the launcher must stay off the GPU stack'
        sys.exit(rc)
    """)
    e = dict(os.environ)
    e.pop("RANK", None)
    e.pop("WORLD_SIZE", None)
    e.update(env or {})
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=e)


def test_ranks_get_their_environment_and_rank0_line_is_relayed(tmp_path):
    body = """
        import json, os, sys
        keys = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"]
        rec = {k: os.environ.get(k) for k in keys}
        rec["argv"] = sys.argv[1:]
        open(os.path.join(os.environ["STUB_DIR"], "rank%s.json" % rec["RANK"]), "w").write(json.dumps(rec))
        print("noise that is not the line")
        if rec["RANK"] == "0":
            print(json.dumps({"metric": "stub", "n_gpus": int(rec["WORLD_SIZE"])}))
        else:
            print(json.dumps({"metric": "not rank 0"}))
    """
    argv = ["--gpus", "3", "--steps", "7", "--warmup", "2", "--config", "C5"]
    p = _run_launcher(tmp_path, 3, argv, body, env={"STUB_DIR": str(tmp_path)})
    assert p.returncode == 0, p.stderr
    out = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(out) == 1 and json.loads(out[0]) == {"metric": "stub", "n_gpus": 3}  # one line, rank 0's
    ports = set()
    for r in range(3):
        rec = json.loads((tmp_path / f"rank{r}.json").read_text())
        assert rec["RANK"] == rec["LOCAL_RANK"] == str(r) and rec["WORLD_SIZE"] == "3"
        assert rec["MASTER_ADDR"] == "127.0.0.1" and rec["argv"] == argv
        assert rec["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ports.add(rec["MASTER_PORT"])
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536


def test_a_failing_rank_fails_the_launch_and_stops_its_peers(tmp_path):
    body = """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)  # a peer stuck in a collective
    """
    p = _run_launcher(tmp_path, 2, ["--gpus", "2"], body)
    assert p.returncode == 7
    assert p.stdout.strip() == ""


def test_no_json_line_is_an_error(tmp_path):
    p = _run_launcher(tmp_path, 2, ["--gpus", "2"], "print('nothing useful')")
    assert p.returncode == 3


def test_main_picks_the_launcher_only_without_a_launcher_environment():
    """--gpus 2 with WORLD_SIZE set (torchrun) must NOT spawn again; a mismatch is an error (rc 2) before any GPU work"""
    e = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=e)
    assert p.returncode == 2 and "WORLD_SIZE=4" in p.stderr
