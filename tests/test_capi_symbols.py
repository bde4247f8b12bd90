"""CPU-side checks of the C-ABI library: it loads, exports every symbol that
include/cuhe_hip.h declares, and its host-only logic (parameter derivation,
per-level helpers: cuhe/Parameters.cu:53-145) matches the golden fixtures.
No compute call is made here (no GPU in this container)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as ge
    from cuhe_amd import build
    if build.stale():
        ge.build()
    from cuhe_amd import capi
    return capi


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "cuhe_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cuhe_hip_\w+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(capi):
    syms = declared_symbols()
    assert len(syms) > 50
    for s in syms:
        assert hasattr(capi.lib, s), "missing export: " + s
        assert s in capi.SIGNATURES, "python binding lacks: " + s
    assert sorted(capi.SIGNATURES) == syms


def test_version_and_error_string(capi):
    assert b"gfx950" in capi.lib.cuhe_hip_version()
    assert capi.lib.cuhe_hip_set_parameters(0, 2, 1, 61, 20, 8191) != 0
    assert b"setParameters" in capi.lib.cuhe_hip_last_error()


@pytest.mark.parametrize("name", ["dhs_simple", "prince", "toy1155", "pow2_16384", "c3_65536", "c4_65536", "c1_pow2_1prime", "c1_prime_m_1prime"])
def test_parameter_derivation(capi, golden, name):
    g = golden("params.json")[name]
    capi.lib.cuhe_hip_reset_parameters()
    capi.check(capi.lib.cuhe_hip_set_parameters(*g["args"]))
    q = capi.get_params()
    for k, v in g["params"].items():
        assert getattr(q, k) == v, k
    for lvl, rec in g["levels"].items():
        assert capi.lib.cuhe_hip_log_coeff(int(lvl)) == rec["logCoeff"]
        assert capi.lib.cuhe_hip_words_coeff(int(lvl)) == rec["wordsCoeff"]
        if "numEvalKey" in rec and q.logRelin:
            assert capi.lib.cuhe_hip_num_eval_key(int(lvl)) == rec["numEvalKey"]
            assert capi.lib.cuhe_hip_num_crt_prime(int(lvl)) == rec["numCrtPrime"]
            assert capi.lib.cuhe_hip_get_level(rec["logCoeff"]) == int(lvl)
    assert capi.lib.cuhe_hip_get_level(q.logMsg) == -1
    capi.lib.cuhe_hip_reset_parameters()


def test_unsupported_ring_is_rejected(capi):
    # phi(m) > 32768 would need a 128K-point cyclic transform (cuhe/Base.cu:59-62 supports 16K/32K/64K only) ...
    assert capi.lib.cuhe_hip_set_parameters(2, 2, 16, 50, 25, 262144) != 0
    assert capi.lib.cuhe_hip_set_parameters(2, 2, 16, 50, 25, 3 * 65536) != 0
    capi.lib.cuhe_hip_reset_parameters()


def test_degree_65536_ring_parameters(capi):
    # ... except x^65536 + 1 (m = 131072), which exists in the negacyclic representation: 64K-point transforms of the
    # full-length residues, primes capped at 23 bits so that 2 n p^2 < P (include/cuhe_hip.h, cuhe_hip_ct_*)
    capi.check(capi.lib.cuhe_hip_set_parameters(25, 2, 16, 552, 23, 131072))
    q = capi.get_params()
    assert (q.modLen, q.crtLen, q.rawLen, q.nttLen) == (65536, 65536, 65536, 65536)
    assert q.logCrtPrime == 23 and q.numCrtPrime == 48 and q.numEvalKey == 69
    assert capi.lib.cuhe_hip_ct_len() == 65536
    capi.lib.cuhe_hip_reset_parameters()


def test_drivers_fail_loudly_without_init(capi):
    assert capi.lib.cuhe_hip_get_crt_primes(None, 0) != 0
    assert b"not initialised" in capi.lib.cuhe_hip_last_error()


@pytest.mark.parametrize("args", [(25, 2, 16, 576, 24, 65536), (3, 2, 8, 40, 20, 1155)])
def test_key_range_covers_the_owned_primes_at_every_level(capi, args):
    """cuhe_hip_key_range (host logic only): the range a participant of the CRT-prime-sharded multiply has to hold contains
    its block of every level (cuhe_hip_shard_bounds), and the ranges of all participants together cost little more than one
    copy of the keys (the evaluation keys are partitioned with the primes)."""
    from cuhe_amd.sharded import shard_bounds
    capi.lib.cuhe_hip_reset_parameters()
    capi.check(capi.lib.cuhe_hip_set_parameters(*args))
    q = capi.get_params()
    f, c, kf, kc = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    for n in range(1, min(9, q.numCrtPrime - q.depth + 2)):
        total = 0
        for r in range(n):
            capi.check(capi.lib.cuhe_hip_key_range(n, r, C.byref(kf), C.byref(kc)))
            total += kc.value
            for lvl in range(q.depth):
                npl = capi.lib.cuhe_hip_num_crt_prime(lvl)
                if npl < n:
                    continue
                capi.check(capi.lib.cuhe_hip_shard_bounds(lvl, n, r, C.byref(f), C.byref(c)))
                assert (f.value, c.value) == shard_bounds(npl, n, r)
                assert kf.value <= f.value and f.value + c.value <= kf.value + kc.value, (n, r, lvl)
        assert q.numCrtPrime <= total <= q.numCrtPrime + (q.depth - 1) * (n - 1), (n, total)
    assert capi.lib.cuhe_hip_key_range(0, 0, C.byref(kf), C.byref(kc)) != 0
    capi.lib.cuhe_hip_reset_parameters()


def test_rccl_symbols_of_comm_hpp_resolve_in_the_image():
    """cuhe_amd/csrc/comm.hpp opens librccl with dlopen and resolves a fixed list of symbols with dlsym: a missing one would
    only show on the first multi-GPU run.  The same list (parsed from the header) is resolved here against the librccl this
    image ships (PyTorch's copy or /opt/rocm's) -- no communicator is made, no GPU is needed."""
    txt = open(os.path.join(ROOT, "cuhe_amd", "csrc", "comm.hpp")).read()
    m = re.search(r"required_symbols\(\)\s*\{.*?names\[\]\s*=\s*\{(.*?)nullptr", txt, flags=re.S)
    assert m, "required_symbols() not found in comm.hpp"
    names = re.findall(r'"(nccl\w+)"', m.group(1))
    used = set(re.findall(r'CUHE_SYM\(\w+,\s*"(nccl\w+)"\)', txt))
    assert set(names) == used and len(names) >= 8, (names, used)
    cands = ["librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"]
    try:
        import torch
        tl = os.path.join(os.path.dirname(torch.__file__), "lib")
        cands = [os.path.join(tl, f) for f in sorted(os.listdir(tl)) if f.startswith("librccl")] + cands
    except ImportError:
        pass
    lib = None
    for c in cands:
        try:
            lib = C.CDLL(c); break
        except OSError:
            continue
    assert lib is not None, "no librccl in this image"
    for n in names + ["ncclCommCount", "ncclCommUserRank", "ncclGetVersion"]:
        assert hasattr(lib, n), "librccl lacks " + n
    ver = C.c_int(0)
    assert lib.ncclGetVersion(C.byref(ver)) == 0 and ver.value > 20000


def test_reference_library_names_exist(capi):
    """The reference's build files produce a static `cuHE` and a shared `cuHEShared` and its examples look for the latter
    (cuhe/CMakeLists.txt:25,38; examples/DHS/CMakeLists.txt:14): both names exist beside libcuHE.so and carry the C++ API."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "cuhe_amd", "cxx"), "-s"])
    lib = os.path.join(ROOT, "cuhe_amd", "lib")
    for name in ("libcuHE.so", "libcuHEShared.so"):
        syms = subprocess.run(["nm", "-D", "-C", "--defined-only", os.path.join(lib, name)], capture_output=True, text=True, check=True).stdout
        assert "cuHE::initCuHE" in syms and "cuHE::setParameters" in syms and "cuHE::multiGPUs" in syms, name
    members = subprocess.run(["ar", "t", os.path.join(lib, "libcuHE.a")], capture_output=True, text=True, check=True).stdout.split()
    assert sorted(members) == ["CuHE.o", "Scheduler.o", "Utils.o"]


def test_exchange_path_policy_for_every_level_of_config4(capi, golden):
    """SURVEY 8(e) / VERDICT r04 item 4: the one exchange of the sharded multiply is an RCCL all-gather.  Host logic of
    cuhe_hip_allgather_rows, without a GPU: for every level of config 4 (both rings) and 1 ... 9 ranks the policy picks ONE in-place
    ncclAllGather when the level's primes split evenly, ONE padded ncclAllGather otherwise, the broadcast group only when a rank
    would own no prime; one rank exchanges nothing unless forced."""
    lib = capi.lib
    for args in ([25, 2, 16, 576, 24, 65536], [25, 2, 16, 552, 23, 131072]):
        lib.cuhe_hip_reset_parameters()
        capi.check(lib.cuhe_hip_set_parameters(*args))
        depth = capi.get_params().depth
        assert depth == 25
        seen = set()
        for lvl in range(depth):
            np_ = lib.cuhe_hip_num_crt_prime(lvl)
            assert lib.cuhe_hip_exchange_path(lvl, 1, 0) == 0
            assert lib.cuhe_hip_exchange_path(lvl, 1, 1) == 1          # forced on one rank: equal blocks by definition
            for nranks in range(2, 10):
                want = 3 if np_ < nranks else (1 if np_ % nranks == 0 else 2)
                assert lib.cuhe_hip_exchange_path(lvl, nranks, 0) == want, (lvl, np_, nranks)
                assert lib.cuhe_hip_exchange_path(lvl, nranks, 2) == 2 and lib.cuhe_hip_exchange_path(lvl, nranks, 3) == 3
                seen.add(want)
                # the blocks the paths move: contiguous, balanced, covering
                import ctypes as C2
                at = 0
                for r in range(nranks):
                    f, c = C2.c_int(), C2.c_int()
                    capi.check(lib.cuhe_hip_shard_bounds(lvl, nranks, r, C2.byref(f), C2.byref(c)))
                    assert f.value == at and c.value in (np_ // nranks, np_ // nranks + 1)
                    at += c.value
                assert at == np_
        assert lib.cuhe_hip_exchange_path(0, 2, 0) == 1 and lib.cuhe_hip_exchange_path(0, 4, 0) == 1 and lib.cuhe_hip_exchange_path(0, 8, 0) == 1
        assert seen >= {1, 2}
        assert lib.cuhe_hip_exchange_path(depth, 2, 0) == -1
    lib.cuhe_hip_reset_parameters()


def test_device_local_cpus_without_a_gpu_changes_nothing(capi):
    """cuhe_hip_device_local_cpus / cuhe_hip_pin_thread_to_device before any HIP call of the process read sysfs only (no compute, no HIP): on a machine
    without an AMD GPU the list is empty and the calling thread keeps its affinity (include/cuhe_hip.h; the parsing of the lists is the same code the GPU
    test exercises, and bench_aux/placement.py's Python twin is tested against a sysfs tree in tests/test_bench_record.py)."""
    import threading
    buf = C.create_string_buffer(256)
    assert capi.lib.cuhe_hip_device_local_cpus(0, buf, 256) == 0
    listed = buf.value.decode()
    assert capi.lib.cuhe_hip_device_local_cpus(-1, buf, 256) != 0          # a bad device is an error, not a crash
    out = {}

    def body():
        tid = threading.get_native_id()
        out["before"] = os.sched_getaffinity(tid)
        out["rc"] = capi.lib.cuhe_hip_pin_thread_to_device(0)
        out["after"] = os.sched_getaffinity(tid)

    th = threading.Thread(target=body); th.start(); th.join()
    assert out["after"] <= out["before"] and len(out["after"]) > 0
    if not listed:
        assert out["rc"] == 0 and out["after"] == out["before"]
