"""The host pool under ThreadSanitizer (CPU): tests/native/thread_pool_stress.cpp forks thousands of times with item counts
that change from fork to fork and lets the workers fall asleep in between; any item run twice / not at all, any data race
and any lost wake-up (a hang) fails."""
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.parametrize("flags", [["-O2"], ["-O1", "-g", "-fsanitize=thread"]])
def test_thread_pool_stress(tmp_path, flags):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = tmp_path / "tp"
    cmd = ["g++", "-std=c++17", "-pthread", *flags, "-I", str(ROOT / "versatiles-glyphs-rs_amd" / "csrc" / "host"),
           str(ROOT / "tests" / "native" / "thread_pool_stress.cpp"), "-o", str(exe)]
    built = subprocess.run(cmd, capture_output=True, text=True)
    if built.returncode != 0 and "-fsanitize=thread" in flags:
        pytest.skip("ThreadSanitizer runtime not available: " + built.stderr[-200:])
    assert built.returncode == 0, built.stderr
    import os
    # default policy (a few workers poll, the others sleep at once), nobody polls, everybody polls
    for spinners in (None, "0", "999", "light"):
        env = dict(os.environ)
        if spinners == "light":   # every third fork is a light one: at most two sleeping workers are woken for it
            env["VG_POOL_LIGHT"] = "1"
        elif spinners is not None:
            env["VG_POOL_SPINNERS"] = spinners
        run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=240, env=env)
        assert run.returncode == 0 and run.stdout.startswith("OK") and "ThreadSanitizer" not in run.stderr, \
            f"spinners={spinners}: " + run.stdout + run.stderr[-2000:]
