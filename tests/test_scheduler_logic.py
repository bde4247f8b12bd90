"""CPU: the gate scheduler of the C++ layer (cuhe_amd/cxx/Scheduler.cpp) without a GPU.  tests/cxx/test_scheduler_logic.cpp compiles
the scheduler's translation unit against a mock of the few C-ABI calls it makes (streams, events, blocks) and records random gate
programs on it: host order, DEVICE order (vector clocks through the lazily recorded per-stream events, across streams and across
devices), batch membership, exactly-once execution, per-device workers, drain / wait / stop / restart -- for 1, 2 and 8 devices and
every batch policy.  Also under ThreadSanitizer when the toolchain has it (the graph has no global lock since round 5)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cxx", "test_scheduler_logic.cpp")
INC = ["-I" + os.path.join(ROOT, "cuhe_amd", "cxx"), "-I" + os.path.join(ROOT, "cuhe_amd", "cxx", "mini_ntl")]


def build(tmp_path, name, extra):
    exe = str(tmp_path / name)
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wno-unused-function"] + extra + INC + ["-o", exe, SRC, "-lpthread"],
                       capture_output=True, text=True, timeout=600)
    return exe, r


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_random_gate_programs_on_the_mock_device(tmp_path):
    exe, r = build(tmp_path, "sched_logic", [])
    assert r.returncode == 0, r.stderr[-3000:]
    run = subprocess.run([exe, "3"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "ALL PASSED" in run.stdout, (run.stdout[-2000:], run.stderr[-2000:])


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_no_data_race_under_thread_sanitizer(tmp_path):
    exe, r = build(tmp_path, "sched_logic_tsan", ["-fsanitize=thread"])
    if r.returncode != 0 and ("tsan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip("this g++ has no ThreadSanitizer runtime")
    assert r.returncode == 0, r.stderr[-3000:]
    # CUHE_SCHED_QUIET_US=0: no timed waits -- the libtsan of gcc 11 does not intercept pthread_cond_clockwait (condition_variable::wait_for),
    # does not see the mutex released during such a wait and reports "double lock" / races on everything the mutex guards
    run = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=900, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0", CUHE_SCHED_QUIET_US="0", CUHE_SCHED_WATCHDOG_S="0"))   # (the watchdog's waits are timed too)
    assert "ALL PASSED" in run.stdout, (run.stdout[-2000:], run.stderr[-2000:])
    assert "ThreadSanitizer" not in run.stderr, run.stderr[-4000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_policy_one_fills_the_groups_of_a_prince_block(tmp_path):
    """the gate program of a homomorphic PRINCE block (19 332 gates recorded S-box by S-box like tests/cxx/test_prince_flow.cpp) with the
    host cost of a gate emulated: the round-5 policy (complete groups first, incomplete ones only when nothing else can progress, oldest
    first) needs several times fewer batch calls than the round-4 policy for the same gates"""
    exe, r = build(tmp_path, "sched_logic_o2", ["-O2"])
    assert r.returncode == 0, r.stderr[-3000:]
    calls = {}
    for pol in (0, 1):
        out = subprocess.run([exe, "prince", str(pol), "3", "1"], capture_output=True, text=True, timeout=600).stdout
        line = [l for l in out.splitlines() if l.startswith("batches:")][0]
        calls[pol] = (int(line.split()[1]), int(line.split("for")[1].split()[0]))
    assert calls[0][1] > 15000 and calls[1][1] > 15000, calls           # (gates that went through batches)
    assert calls[1][0] * 2 < calls[0][0], calls
