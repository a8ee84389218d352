"""bench.py's launch plumbing on CPU: `--gpus N` started plainly spawns N ranks (rendezvous on 127.0.0.1, barrier and
max-over-ranks timing over gloo, ONE JSON line from rank 0), and a rank that cannot get its GPU fails the whole run."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *argv], capture_output=True, text=True, timeout=300, env=e)


def test_gpus_flag_spawns_ranks_dry_run():
    for workload in ("single", "crt"):
        r = _run("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--workload", workload)
        assert r.returncode == 0, r.stderr
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout                      # rank 0 only, nothing else on stdout
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["dry_run"] is True
        assert out["value"] is None and out["vs_baseline"] is None and out["workload"] == workload
    r = _run("--dry-run", "--steps", "2", "--warmup", "0")
    assert json.loads(r.stdout)["n_gpus"] == 1


def test_rank_env_is_honoured_like_torchrun(tmp_path):
    """Under torch.distributed.run the ranks arrive through the environment; --gpus then spawns nothing."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    json_lines = [[ln for ln in o[0].splitlines() if ln.startswith("{")] for o in outs]
    assert len(json_lines[0]) == 1 and json_lines[1] == []
    assert json.loads(json_lines[0][0])["n_gpus"] == 2


def test_without_gpus_the_bench_fails_loudly():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu")
    assert r.returncode != 0 and "error" in r.stdout
    r = _run("--steps", "1", "--warmup", "0", "--no-cpu")
    assert r.returncode != 0                                  # no CPU fallback: the product needs its GPU


def test_recorded_counters_are_refused_when_kernels_changed(tmp_path, monkeypatch):
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    tag = bench.kernel_source_tag()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_tag", lambda: tag)
    rec = {"workloads": {"single_b256": {"kernels": {"row_mul": 123.0}}}}
    (prof / "traffic.json").write_text(json.dumps(dict(rec, kernel_source_tag=tag)))
    assert bench.recorded_counters("traffic", "single_b256", "row_mul") == (123.0, "current")
    assert bench.recorded_counters("traffic", "single_b256")[0] == {"kernels": {"row_mul": 123.0}}
    val, state = bench.recorded_counters("traffic", "crt_b32_k4", "row_mul")
    assert val is None and state == "not recorded for this workload"
    (prof / "traffic.json").write_text(json.dumps(dict(rec, kernel_source_tag="0" * 16)))
    val, state = bench.recorded_counters("traffic", "single_b256", "row_mul")
    assert val is None and state.startswith("stale")
    assert bench.recorded_counters("valu", "single_b256", "row_mul") == (None, "absent")


def test_bound_follows_the_counter_records(tmp_path, monkeypatch):
    """roofline.bound names the VALU when the issue model fills more of the dominant kernel's time than its HBM bytes do."""
    spec = importlib.util.spec_from_file_location("bench_under_test3", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    tag = bench.kernel_source_tag()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_tag", lambda: tag)
    kinds = ("col_fwd", "row_mul", "col_inv")
    (prof / "traffic.json").write_text(json.dumps({"kernel_source_tag": tag, "workloads": {"w": {"kernels": {k: 8.0e8 for k in kinds}}}}))
    (prof / "valu.json").write_text(json.dumps({"kernel_source_tag": tag, "workloads": {"w": {
        "kernels": {"col_fwd": 8.0e7, "row_mul": 9.0e7, "col_inv": 5.0e7}, "clock_GHz": {k: 2.0 for k in kinds}}}}))
    timed = {"col_fwd": (0.18 * 5, 5), "row_mul": (0.2 * 5, 5), "col_inv": (0.12 * 5, 5)}
    roof, _ = bench.roofline_of(timed, {"col_fwd": 8.05e8, "row_mul": 8.05e8, "col_inv": 5.4e8}, {k: 1 for k in kinds}, "w")
    busy = 9.0e7 * 4.6 / (1024 * 2.0e9 * 0.2e-3)
    assert roof["kernel"] == "row_mul" and abs(roof["valu"]["issue_busy_frac"] - busy) < 1e-9
    assert roof["bound"] == "valu" and roof["frac"] < busy and roof["traffic"] == 8.0e8
    assert 0.9 < roof["valu"]["whole_step"]["issue_busy_frac"] < 1.0
    (prof / "valu.json").unlink()
    roof, _ = bench.roofline_of(timed, {"col_fwd": 8.05e8, "row_mul": 8.05e8, "col_inv": 5.4e8}, {k: 1 for k in kinds}, "w")
    assert roof["bound"] == "hbm"


def test_roofline_arithmetic():
    spec = importlib.util.spec_from_file_location("bench_under_test2", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    prof = {"col_fwd": (2.0, 2), "row_mul": (3.0, 2), "col_inv": (1.0, 2), "crt": (0.0, 0)}
    roof, per = bench.roofline_of(prof, {"col_fwd": 12e6, "row_mul": 12e6, "col_inv": 8e6}, {"col_fwd": 1, "row_mul": 1, "col_inv": 1}, None)
    assert roof["kernel"] == "row_mul" and abs(roof["achieved"] - 12e6 / 1.5e-3 / 1e9) < 1e-9
    assert abs(roof["frac"] - roof["achieved"] / 8000.0) < 1e-12
    assert abs(roof["whole_step"]["achieved"] - 32e6 / 3e-3 / 1e9) < 1e-9
    assert per == {"col_fwd": 1.0, "row_mul": 1.5, "col_inv": 0.5}
