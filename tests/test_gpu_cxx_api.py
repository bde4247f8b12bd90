"""-m gpu: builds and runs the C++ drop-in API test (tests/cxx/test_cuhe_api.cpp): CuHE.h classes and
gates on top of the C ABI, checked against host ZZX arithmetic (NTL if installed, else mini_ntl)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sched", [False, True], ids=["synchronous", "scheduled"])
def test_cxx_api_program(sched):
    """(scheduled: the same program with CUHE_SCHED=1 in the environment -- every CuCtxt gate and conversion goes through
    the gate scheduler of cuhe_amd/cxx/Scheduler.h, CUHE_SCHED_CHECK=1 verifies the client-side metadata mirror)"""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    import __graft_entry__ as ge
    ge.build()
    cxx = os.path.join(ROOT, "cuhe_amd", "cxx")
    subprocess.check_call(["make", "-C", cxx, "-s", "test"])
    exe = os.path.join(ROOT, "cuhe_amd", "lib", "test_cuhe_api")
    env = dict(os.environ, CUHE_SCHED="1", CUHE_SCHED_CHECK="1") if sched else dict(os.environ, CUHE_SCHED="0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ALL PASSED" in r.stdout


@pytest.mark.parametrize("sched", [False, True], ids=["synchronous", "scheduled"])
@pytest.mark.parametrize("params", [(3, 2, 8, 40, 20, 1155), (5, 2, 1, 61, 20, 8191), (3, 2, 16, 48, 24, 32768),
                                    (3, 2, 16, 50, 25, 32767), (3, 2, 16, 50, 25, 32749)],
                         ids=["toy1155", "dhs_simple", "pow2_32768-negacyclic", "phi32767-generic-64K", "prime32749-fold-64K"])
def test_dhs_scheme_flow(params, sched):
    """keygen / encrypt / XOR / NOT / AND + relin + modSwitch (two levels) / decrypt through CuHE.h -- the checks of
    examples/DHS/simple_DHS.cu:49-170, with the reference example's own parameter set as the second case"""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    import __graft_entry__ as ge
    ge.build()
    cxx = os.path.join(ROOT, "cuhe_amd", "cxx")
    subprocess.check_call(["make", "-C", cxx, "-s", "test"])
    exe = os.path.join(ROOT, "cuhe_amd", "lib", "test_dhs_flow")
    env = dict(os.environ, CUHE_SCHED="1", CUHE_SCHED_CHECK="1") if sched else dict(os.environ, CUHE_SCHED="0")
    r = subprocess.run([exe] + [str(v) for v in params], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ALL PASSED" in r.stdout and "wrong" not in r.stdout
    # checkKeys (examples/DHS/simple_DHS.cu:165-205): second / third scheme objects from key strings in the same process, then the first again
    assert "keys\tright" in r.stdout and "and after re-initialisation\tright" in r.stdout and "generation unchanged" in r.stdout


@pytest.mark.parametrize("flags", [["--threads", "8"], ["--threads", "4", "--async"], ["--threads", "1", "--async"],
                                   ["--threads", "6", "--async", "--devices", "3", "--virtual"],
                                   ["--threads", "1", "--sched"], ["--threads", "1", "--sched", "3", "--devices", "3", "--virtual"],
                                   ["--threads", "1", "--sched", "5", "--no-batching"], ["--threads", "1", "--default"], ["--threads", "1", "--sched", "1"],
                                   ["--threads", "8", "--zzx-state", "--default"], ["--threads", "1", "--zzx-state", "--default"], ["--threads", "8", "--zzx-state", "--no-round-checks"],
                                   ["--threads", "4", "--zzx-state", "--default", "--devices", "3", "--virtual", "--no-round-checks"]],
                         ids=["sync-8-threads", "async-4-threads", "async-1-thread", "async-3-virtual-devices",
                              "scheduled-1-thread", "scheduled-1-thread-3-virtual-devices", "scheduled-1-thread-no-batching",
                              "library-default-1-thread", "scheduled-1-thread-1-worker",
                              "zzx-state-library-default-8-threads", "zzx-state-library-default-1-thread", "zzx-state-sync-8-threads",
                              "zzx-state-library-default-3-virtual-devices"])
def test_prince_known_answer(flags):
    """BASELINE config 5 on one GPU: homomorphic PRINCE through CuHE.h (tests/cxx/test_prince_flow.cpp).  The
    reference's known answer 0x9fb51935fc3df524 (examples/Prince/Prince.cu:96) and its 12 intermediate round states
    (Prince.cu:108-145) must decrypt bit for bit; 1920 cAnd / 1152 relin / depth 24 as in the reference's circuit.
    Run with the reference's synchronous gate semantics on 8 host threads / streams, and with asynchronous gates
    (setAsynchronous: stream-ordered buffers, one synchronisation per S-box) on 4 threads and on the default stream;
    and in the reference's multi-GPU arrangement (Prince.cu:194-200: the S-boxes of a layer spread over the devices'
    threads, moveTo to and from the device that holds the state) on three virtual devices of the one GPU.
    scheduled-*: ONE client thread, default stream, a gate per call (the reference client's pattern) with the library's gate
    scheduler on (setScheduled): ready gates of one kind run as one call of the array entry points; CUHE_SCHED_CHECK=1
    verifies the client-side metadata mirrors whenever an object is taken back.
    zzx-state-*: the reference example's LITERAL structure (Prince.cu:188-322): the state lives on the host as ZZX between S-boxes, every
    S-box hands four ZZX in (setLevel(lvl, dev, ZZX)) and takes four back (x2z ; zRep), the linear layers are host additions."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    import __graft_entry__ as ge
    ge.build()
    cxx = os.path.join(ROOT, "cuhe_amd", "cxx")
    subprocess.check_call(["make", "-C", cxx, "-s", "test"])
    exe = os.path.join(ROOT, "cuhe_amd", "lib", "test_prince_flow")
    env = dict(os.environ, CUHE_SCHED_CHECK="1")
    env.pop("CUHE_SCHED", None)
    if "--no-batching" in flags:                    # every recorded gate runs its own closure (CUHE_SCHED_BATCH=0): the path the batches replace
        env["CUHE_SCHED_BATCH"] = "0"
        flags = [f for f in flags if f != "--no-batching"]
    r = subprocess.run([exe] + flags, capture_output=True, text=True, timeout=1200, env=env)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ALL PASSED" in r.stdout and "wrong" not in r.stdout
    assert "homomorphic PRINCE: 9fb51935fc3df524" in r.stdout
    assert r.stdout.count("right") == (1 if "--no-round-checks" in flags else 13)
    if "--sched" in flags or "--default" in flags:    # --default: no setScheduled call and no environment variable -- what an unchanged reference client gets
        assert "scheduled gates" in r.stdout
    else:
        assert "scheduled gates" not in r.stdout


def test_prince_known_answer_on_arrays():
    """The same circuit evaluated layer by layer with the gates on ARRAYS of ciphertexts of include/cuhe_hip.h
    (tests/cxx/test_prince_batched.cpp: ~12 calls per S-box layer over 64-224 ciphertexts each).  Same known answer and
    round states as test_prince_known_answer."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    import __graft_entry__ as ge
    ge.build()
    cxx = os.path.join(ROOT, "cuhe_amd", "cxx")
    subprocess.check_call(["make", "-C", cxx, "-s", "test"])
    exe = os.path.join(ROOT, "cuhe_amd", "lib", "test_prince_batched")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=1200)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ALL PASSED" in r.stdout and "wrong" not in r.stdout
    assert "homomorphic PRINCE: 9fb51935fc3df524" in r.stdout
    assert r.stdout.count("right") == 13


def test_in_process_multi_device_on_virtual_devices():
    """multiGPUs(3) on one physical GPU (cuhe_hip_set_virtual_devices): mulZZX on every device, moveTo / copyTo,
    cAnd + relin with the keys resident on another device, one host thread per device (cuhe/CuHE.cu:217-256,
    examples/Prince/Prince.cu:194-200)"""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    import __graft_entry__ as ge
    ge.build()
    cxx = os.path.join(ROOT, "cuhe_amd", "cxx")
    subprocess.check_call(["make", "-C", cxx, "-s", "test"])
    exe = os.path.join(ROOT, "cuhe_amd", "lib", "test_multi_device")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ALL PASSED" in r.stdout


@pytest.mark.parametrize("flags", [[], ["--async"], ["--devices", "3", "--virtual"], ["--devices", "8", "--virtual", "--async", "--no-round-checks"]],
                         ids=["sync", "async", "3-virtual-devices", "8-virtual-devices-async"])
def test_prince_known_answer_on_cxx_array_classes(flags):
    """The same circuit through the C++ array classes of cuhe_amd/cxx/CuHEArray.h (CuCtxtArray / CuIndexTable: cAnd over
    index pairs, cXor over index lists, relin / modSwitch / x2n / x2c on whole arrays, copy / concat): known answer and
    the 12 round states, with the reference's synchronise-per-operation semantics and with asynchronous gates."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    import __graft_entry__ as ge
    ge.build()
    cxx = os.path.join(ROOT, "cuhe_amd", "cxx")
    subprocess.check_call(["make", "-C", cxx, "-s", "test"])
    exe = os.path.join(ROOT, "cuhe_amd", "lib", "test_prince_arrays_cxx")
    r = subprocess.run([exe] + flags, capture_output=True, text=True, timeout=1200)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ALL PASSED" in r.stdout and "wrong" not in r.stdout
    assert "homomorphic PRINCE: 9fb51935fc3df524" in r.stdout
    assert r.stdout.count("right") == (1 if "--no-round-checks" in flags else 13)


def _soak_exe():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    import __graft_entry__ as ge
    ge.build()
    cxx = os.path.join(ROOT, "cuhe_amd", "cxx")
    subprocess.check_call(["make", "-C", cxx, "-s", "test"])
    return os.path.join(ROOT, "cuhe_amd", "lib", "test_sched_soak")


@pytest.mark.parametrize("mode", ["default", "CUHE_SCHED=0", "CUHE_SCHED_THREADS=1"])
def test_scheduler_soak_conditions(mode):
    """tests/cxx/test_sched_soak.cpp: operands destroyed while their gates are queued, setScheduled(false) with work queued and a restart,
    a raw pointer read right after a gate was recorded (the reference's observable synchronous semantics, cuhe/CuHE.cu:98,121,139,157),
    two client threads, a second initCuHE on the same ring with gates queued -- on the library's default (scheduled gates since round 6:
    no environment variable, no setScheduled call), on synchronous gates, and with ONE worker per device."""
    exe = _soak_exe()
    env = dict(os.environ, CUHE_SCHED_CHECK="1")
    env.pop("CUHE_SCHED", None)
    if "=" in mode:
        k, v = mode.split("=")
        env[k] = v
    r = subprocess.run([exe, "2"], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0 and "ALL PASSED" in r.stdout and "wrong" not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert ("gates are synchronous" if mode == "CUHE_SCHED=0" else "gates are scheduled") in r.stdout
    assert "watchdog" not in r.stderr


@pytest.mark.parametrize("sched", ["default", "CUHE_SCHED=0"])
@pytest.mark.parametrize("n", [0, 3, 40, 150])
def test_allocation_failure_ends_the_program_like_the_reference(n, sched):
    """failure injection (cuhe_hip_set_alloc_fail_after): the (n+1)-th device allocation fails -- inside a recorded gate, inside a batch
    of the scheduler, or on the client thread.  The reference's convention for a failed cudaMalloc is CSC: a message and exit(-1)
    (cuhe/Debug.h:35-66); the program must end that way within seconds, on scheduled and on synchronous gates: no hang, no crash."""
    import time
    exe = _soak_exe()
    env = dict(os.environ)
    env.pop("CUHE_SCHED", None)
    if sched != "default":
        env["CUHE_SCHED"] = "0"
    t0 = time.time()
    r = subprocess.run([exe, "allocfail", str(n)], capture_output=True, text=True, timeout=300, env=env)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 255, (r.returncode, r.stdout[-1000:], r.stderr[-1000:])          # exit(-1)
    assert "cuheSafeCall() failed" in r.stderr and "injected" in r.stderr
    assert time.time() - t0 < 120


def test_worker_threads_run_on_the_cpus_local_to_their_device():
    """cuhe_hip_device_local_cpus / cuhe_hip_pin_thread_to_device (include/cuhe_hip.h): the kernel's local_cpulist of the GPU's PCI function, and a
    thread narrowed to it (what the gate scheduler does with its workers: blocks issued from the far socket of a two-socket host take 0.065-0.070 s
    against 0.058-0.059 s, profiles/r06_numa_pinning.txt).  Checked in a thread of its own: the affinity of the pytest process is left alone."""
    import ctypes
    import threading
    import torch
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from cuhe_amd import capi
    buf = ctypes.create_string_buffer(1024)
    assert capi.lib.cuhe_hip_device_local_cpus(0, buf, 1024) == 0
    cpulist = buf.value.decode()
    print("local cpulist of device 0:", repr(cpulist))
    out = {}

    def body():
        tid = threading.get_native_id()
        before = os.sched_getaffinity(tid)
        out["rc"] = capi.lib.cuhe_hip_pin_thread_to_device(0)
        out["before"], out["after"] = before, os.sched_getaffinity(tid)

    th = threading.Thread(target=body); th.start(); th.join()
    assert out["after"] <= out["before"] and len(out["after"]) > 0
    if not cpulist:
        assert out["rc"] == 0 and out["after"] == out["before"]          # the platform does not say: nothing changes
        return
    local = set()
    for part in cpulist.split(","):
        a, _, b = part.partition("-")
        local.update(range(int(a), int(b or a) + 1))
    want = out["before"] & local
    if want and want != out["before"]:
        assert out["rc"] == 1 and out["after"] == want
    else:
        assert out["rc"] == 0 and out["after"] == out["before"]
    assert os.sched_getaffinity(threading.main_thread().native_id) == out["before"]       # the main thread (same mask before the call) was not touched
    # the thread that brings the library up is narrowed only WHILE it does so (CUHE_PIN_CLIENT unset = 2): same mask after the call, whatever it returns
    seen = {}

    def bring_up():
        tid = threading.get_native_id()
        seen["before"] = os.sched_getaffinity(tid)
        seen["rc"] = capi.lib.cuhe_hip_multi_gpus(1)          # (refused when another test has initialised the library: the placement code runs first either way)
        seen["after"] = os.sched_getaffinity(tid)

    th = threading.Thread(target=bring_up); th.start(); th.join()
    assert seen["after"] == seen["before"]
