"""CPU: `python3 bench.py --gpus N` starts its N ranks by itself (BASELINE.json's metric is quoted "at 1/2/4/8 GPU",
SURVEY §8(e)).  The launcher runs before anything touches the GPU, relays rank 0's JSON line as the only line of stdout, returns the
worst child's exit code and refuses a world it cannot fill."""
import json
import os
import subprocess
import sys
import time
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env_without_launcher():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_dry_launch_shows_one_process_per_gpu():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "7", "--warmup", "3", "--dry-launch"], capture_output=True, text=True,
                       env=_env_without_launcher(), timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1
    plan = json.loads(lines[0])["dry_launch"]
    assert [p["rank"] for p in plan] == list(range(8))
    ports = {p["env"]["MASTER_PORT"] for p in plan}
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536
    for r_, p in enumerate(plan):
        e = p["env"]
        assert e["RANK"] == e["LOCAL_RANK"] == str(r_) and e["WORLD_SIZE"] == "8" and e["MASTER_ADDR"] == "127.0.0.1"
        assert p["argv"][1] == BENCH and p["argv"][2:] == ["--gpus", "8", "--steps", "7", "--warmup", "3"]   # same arguments, no --dry-launch


def test_a_world_that_cannot_be_filled_is_refused_not_shrunk():
    """No GPU in this container: --gpus 2 must end non-zero with a one-line reason and WITHOUT a result line (never a 1-GPU number)."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU box: covered by the GPU tests")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=_env_without_launcher(), timeout=600)
    assert r.returncode == 2 and r.stdout == ""
    assert r.stderr.strip().splitlines()[-1].startswith("bench.py: --gpus 2 but 0 GPU(s) visible")


def test_a_launcher_that_started_another_world_is_refused():
    env = _env_without_launcher()
    env.update(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "1"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and r.stdout == "" and "WORLD_SIZE=4 but --gpus 8" in r.stderr


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _fake(script):
    return [sys.executable, "-c", script]


def test_relay_prints_rank0_line_last_and_only(capfd, monkeypatch):
    b = _bench_module()
    line = json.dumps({"metric": "m", "value": 1.0, "n_gpus": 3})
    scripts = ["import os,sys,time; print('banner'); print(%r); sys.stdout.flush(); time.sleep(0.3); print('late noise from rank 0')" % line,
               "import os; print('hello from', os.environ['RANK'], os.environ['WORLD_SIZE'])",
               "import os,time; time.sleep(0.5); print('{\"metric\": \"not rank 0\"}')"]
    monkeypatch.setattr(b, "rank_commands", lambda n, argv, port: [(_fake(s), {"RANK": str(r), "WORLD_SIZE": str(n)}) for r, s in enumerate(scripts)])
    monkeypatch.setattr(b, "visible_gpus", lambda: 3)
    code = b.launch_ranks(types.SimpleNamespace(gpus=3, dry_launch=False), [])
    out, err = capfd.readouterr()
    assert code == 0 and out == line + "\n"
    assert "[rank 0] banner" in err and "[rank 0] late noise from rank 0" in err and "[rank 1] hello from 1 3" in err and "not rank 0" in err


def test_a_failing_rank_stops_the_others_and_sets_the_exit_code(capfd, monkeypatch):
    b = _bench_module()
    scripts = ["import time; time.sleep(600)", "import sys; sys.exit(7)", "import time; time.sleep(600)"]
    monkeypatch.setattr(b, "rank_commands", lambda n, argv, port: [(_fake(s), {}) for s in scripts])
    monkeypatch.setattr(b, "visible_gpus", lambda: None)      # count unknown: the ranks are started and fail by themselves
    t0 = time.time()
    code = b.launch_ranks(types.SimpleNamespace(gpus=3, dry_launch=False), [])
    out, err = capfd.readouterr()
    assert code == 7 and out == "" and time.time() - t0 < 60      # the failing rank's own code; the two ranks stopped from the launcher do not count
    assert "rank 1 exited with code 7" in err


def test_no_result_line_is_an_error(capfd, monkeypatch):
    b = _bench_module()
    monkeypatch.setattr(b, "rank_commands", lambda n, argv, port: [(_fake("print('nothing useful')"), {}) for _ in range(n)])
    monkeypatch.setattr(b, "visible_gpus", lambda: 2)
    assert b.launch_ranks(types.SimpleNamespace(gpus=2, dry_launch=False), []) == 1
    out, _ = capfd.readouterr()
    assert out == ""


def test_ranks_that_never_end_are_stopped_after_the_launch_timeout_and_killed_if_they_ignore_it(capfd, monkeypatch):
    """a rank stuck in a collective that ignores SIGTERM must not hang the launcher — overall limit, then SIGKILL ten seconds after SIGTERM."""
    b = _bench_module()
    scripts = ["import signal,time; signal.signal(signal.SIGTERM, signal.SIG_IGN); print('deaf', flush=True); time.sleep(600)", "import time; time.sleep(600)"]
    monkeypatch.setattr(b, "rank_commands", lambda n, argv, port: [(_fake(s), {}) for s in scripts])
    monkeypatch.setattr(b, "visible_gpus", lambda: 2)
    t0 = time.time()
    code = b.launch_ranks(types.SimpleNamespace(gpus=2, dry_launch=False, launch_timeout=2.0), [])
    out, err = capfd.readouterr()
    assert code == 124 and out == "" and 11 < time.time() - t0 < 40
    assert "still running after --launch-timeout 2 s" in err
