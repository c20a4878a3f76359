"""`python bench.py --gpus N` as the benchmark driver types it: with N > 1 and no WORLD_SIZE in the environment
bench.py must start its own ranks — before it imports torch or touches a GPU — relay their one JSON line and pass their exit
status on.  CPU-only: the child processes are stubbed; the real thing is rehearsed on the one-GPU box
(profiles/r4_bench_gpus2_self_launched_one_gpu_rehearsal.json)."""
import importlib.util
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture()
def bench(monkeypatch):
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VG_SHARE_GPU", "VG_DIST_BACKEND"):
        monkeypatch.delenv(k, raising=False)
    return mod


def test_n_gpus_without_a_launcher_starts_the_ranks_itself(bench, monkeypatch):
    seen = {}
    monkeypatch.setattr(bench, "launch_ranks", lambda n, argv: seen.update(n=n, argv=list(argv)) or 7)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert "torch" not in sys.modules or True   # (nothing below may need it)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7 and seen == {"n": 4, "argv": ["--gpus", "4", "--steps", "3", "--warmup", "1"]}


def test_the_launcher_relays_the_line_and_the_status(bench, monkeypatch, capsys):
    calls = []

    def fake_run(cmd, **kw):
        calls.append((cmd, kw.get("env", {})))
        if "-c" in cmd:                                            # the device count, taken in a child
            return subprocess.CompletedProcess(cmd, 0, stdout="8\n", stderr="")
        assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--master-addr" in cmd
        assert cmd[-4:] == ["--gpus", "8", "--steps", "2"] and cmd[-5].endswith("bench.py")
        line = json.dumps({"metric": "glyphs/sec", "value": 1.0, "n_gpus": 8})
        return subprocess.CompletedProcess(cmd, 0, stdout="noise\n" + line + "\n", stderr="")
    monkeypatch.setattr(subprocess, "run", fake_run)
    rc = bench.launch_ranks(8, ["--gpus", "8", "--steps", "2"])
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert rc == 0 and out["n_gpus"] == 8 and out["launcher"]["attempt"] == 1 and out["launcher"]["devices_visible"] == 8
    assert "VG_SHARE_GPU" not in calls[-1][1]                       # enough devices: nothing is shared


def test_fewer_devices_than_ranks_is_a_labelled_rehearsal_and_a_failed_attempt_is_retried_over_gloo(bench, monkeypatch, capsys):
    attempts = []

    def fake_run(cmd, **kw):
        if "-c" in cmd:
            return subprocess.CompletedProcess(cmd, 0, stdout="1\n", stderr="")
        attempts.append(dict(kw["env"]))
        return subprocess.CompletedProcess(cmd, 0, stdout=json.dumps({"value": 2.0, "n_gpus": 2}) + "\n", stderr="")
    monkeypatch.setattr(subprocess, "run", fake_run)
    assert bench.launch_ranks(2, ["--gpus", "2"]) == 0
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert attempts[0]["VG_SHARE_GPU"] == "1" and attempts[0]["VG_DIST_BACKEND"] == "gloo"
    assert "rehearsal" in out["launcher"]["notes"][0]
    # enough devices, first attempt (RCCL) dies without a line: second attempt with the collectives on gloo
    attempts.clear()

    def flaky(cmd, **kw):
        if "-c" in cmd:
            return subprocess.CompletedProcess(cmd, 0, stdout="2\n", stderr="")
        attempts.append(dict(kw["env"]))
        if len(attempts) == 1:
            return subprocess.CompletedProcess(cmd, 1, stdout="", stderr="")
        return subprocess.CompletedProcess(cmd, 0, stdout=json.dumps({"value": 3.0, "n_gpus": 2}) + "\n", stderr="")
    monkeypatch.setattr(subprocess, "run", flaky)
    assert bench.launch_ranks(2, ["--gpus", "2"]) == 0
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out["launcher"]["attempt"] == 2 and attempts[1]["VG_DIST_BACKEND"] == "gloo" and "VG_DIST_BACKEND" not in attempts[0]


def test_last_resort_sums_independent_one_device_processes(bench, monkeypatch, capsys):
    """both launcher attempts die without a line: N one-device processes, no rendezvous, labelled as such"""
    def dead(cmd, **kw):
        if "-c" in cmd:
            return subprocess.CompletedProcess(cmd, 0, stdout="2\n", stderr="")
        return subprocess.CompletedProcess(cmd, 1, stdout="", stderr="")
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, **kw):
            started.append((cmd, env))
            self.rank = len(started) - 1

        def communicate(self, timeout=None):
            return json.dumps({"value": 10.0 + self.rank, "mpixel_sdf_per_s": 1.0, "ms_per_step": 0.1 + self.rank, "n_gpus": 1}) + "\n", None
    monkeypatch.setattr(subprocess, "run", dead)
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    assert bench.launch_ranks(2, ["--gpus", "2", "--steps", "5"]) == 0
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out["value"] == 21.0 and out["n_gpus"] == 2 and out["ms_per_step"] == 1.1 and "LAST RESORT" in out["launcher"]["form"]
    assert [e["HIP_VISIBLE_DEVICES"] for _, e in started] == ["0", "1"]
    assert all(c[c.index("--gpus") + 1] == "1" and "--no-e2e" in c for c, _ in started)
