"""SURVEY.md section 5's sanitizer row: the CPU oracle (the checker) and the product's host-side map mirror under
-fsanitize=address,undefined with every report fatal (`make -C oracle asan`: oracle/asan_oracle.c drives the oracle through its whole
surface, oracle/asan_host_mirror.cpp fuzzes ct_icp_amd/csrc/ctgn_map.hpp against a std::map model), and the product's helper-thread
pool of the host-side staging loops (ct_icp_amd/csrc/ctgn_hostpool.hpp) under the thread sanitizer and under address + undefined
(oracle/tsan_hostpool.cpp: every part exactly once, nothing touched after run() returns). CPU only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def asan_build():
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return os.path.join(ROOT, "oracle", "_build")


@pytest.mark.parametrize("program", ["asan_oracle", "asan_host_mirror"])
def test_sanitized_program_runs_clean(asan_build, program):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="4")
    r = subprocess.run([os.path.join(asan_build, program)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert f"{program} ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


@pytest.mark.parametrize("program", ["tsan_hostpool", "asan_hostpool"])
def test_host_pool_runs_clean_under_sanitizers(asan_build, program):
    import shutil
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1")
    cmd = [os.path.join(asan_build, program)]
    if program.startswith("tsan") and shutil.which("setarch"):
        cmd = ["setarch", "x86_64", "-R"] + cmd          # the thread sanitizer's shadow mapping can collide with a randomised layout
    r = None
    for _ in range(3):                                   # ... which shows as a DEADLYSIGNAL before main(): not a finding, try again
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
        if "DEADLYSIGNAL" not in r.stderr:
            break
    if "DEADLYSIGNAL" in r.stderr and "hostpool" not in r.stdout and "WARNING" not in r.stderr:
        pytest.skip("the thread sanitizer cannot set up its shadow memory in this environment")
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "hostpool ok" in r.stdout and "ThreadSanitizer" not in r.stderr and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
