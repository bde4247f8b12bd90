"""CPU: the bookkeeping between bench.py and the committed counter record (profiles/traffic_rNN.json).

bench.py reports `roofline.traffic` (HBM-side bytes of the dominant kernel, from rocprofv3 PMC passes) only when the record was
measured on the kernel sources it is running: both carry a hash of the CODE of the three kernel headers -- comments and blank
space removed, so that editing a comment does not orphan a measurement, while any change of an instruction does."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_hash_ignores_comments_and_spacing_but_not_code():
    import bench
    a = "int f(int x) {   // adds one\n    return x + 1;  /* really */\n}\n\n"
    b = "int f(int x) {\n\treturn x + 1;\n}\n// trailing remark\n"
    c = "int f(int x) {\n    return x + 2;\n}\n"
    assert bench.code_only(a) == bench.code_only(b)
    assert bench.code_only(a) != bench.code_only(c)
    assert "/*" not in bench.code_only("x = 1; /* a\nmulti-line\nremark */ y = 2;") and "y = 2;" in bench.code_only("x = 1; /* a\nb */ y = 2;")


def test_newest_traffic_record_matches_the_kernel_sources_in_the_tree():
    """the record the round's bench line quotes was measured on THIS code (otherwise bench.py prints traffic: null and says why)"""
    import bench
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")))
    assert recs, "no counter record committed"
    rec = json.load(open(recs[-1]))
    if rec["kernel_sha16"] != bench.kernel_sha16():        # mid-round state: bench.py then reports traffic: null with the reason
        pytest.skip("the kernel headers changed since %s was measured: re-run tools/profile_final.sh before the round ends" % os.path.basename(recs[-1]))
    assert rec["transform_len"] == 65536 and rec["bytes_per_transform"] >= 10 * 65536          # at least the algorithmic bytes
    lanes = sum(rec["valu_lane_instructions_per_point"].values())
    assert 100 < lanes < 250                                                                   # vector lane-instructions per point


def test_live_pmc_record_is_read_back_from_a_rocpd_database(tmp_path, monkeypatch):
    """bench.measure_traffic_live (round 5: the PMC passes run as child processes of bench.py itself) against a stand-in for rocprofv3:
    a script that writes the rocpd tables the function queries.  Checks the query, the units (KB per dispatch, FETCH_SIZE x 2, SQ_INSTS_VALU per
    shader engine x 64 lanes) and the fall-back to None when a pass fails."""
    import sqlite3
    import subprocess
    import bench
    B, L = 8192, 65536
    per_counter = {"FETCH_SIZE": 800000.0, "WRITE_SIZE": 4850000.0, "SQ_INSTS_VALU": 40188928.0}

    def fake_run(cmd, **kw):
        counter, outdir = cmd[cmd.index("--pmc") + 1], cmd[cmd.index("-d") + 1]
        os.makedirs(outdir, exist_ok=True)
        db = sqlite3.connect(os.path.join(outdir, "p_results.db"))
        db.executescript("create table rocpd_info_kernel_symbol(id integer, display_name text); create table rocpd_kernel_dispatch(event_id integer, kernel_id integer);"
                         "create table rocpd_info_pmc(id integer, name text); create table rocpd_pmc_event(event_id integer, pmc_id integer, value real);")
        db.execute("insert into rocpd_info_kernel_symbol values (1, 'void cuhe::ntt_onewg_stream<15, 0, 0>(void*, unsigned int const*)')")
        db.execute("insert into rocpd_info_kernel_symbol values (2, 'void other_kernel(int)')")
        db.execute("insert into rocpd_info_pmc values (7, ?)", (counter,))
        for ev in range(3):
            db.execute("insert into rocpd_kernel_dispatch values (?, 1)", (ev,))
            db.execute("insert into rocpd_pmc_event values (?, 7, ?)", (ev, per_counter[counter]))
        db.execute("insert into rocpd_kernel_dispatch values (9, 2)"); db.execute("insert into rocpd_pmc_event values (9, 7, 1.0)")
        db.commit(); db.close()
        return subprocess.CompletedProcess(cmd, 0, "", "")

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(bench.os.path, "exists", lambda p, _e=os.path.exists: True if p.endswith("rocprofv3") else _e(p))
    rec = bench.measure_traffic_live(L, B, 0)
    assert rec and rec["one_launch"] and rec["dispatches_sampled"] == 3 and rec["transforms_per_launch"] == B
    assert rec["bytes_per_launch"] == int(1024 * (2 * 800000.0 + 4850000.0))
    assert rec["valu_lane_instructions_per_transform"] == int(40188928.0 * 32 * 64 / B)
    monkeypatch.setattr(bench.subprocess, "run", lambda cmd, **kw: subprocess.CompletedProcess(cmd, 1, "", "boom"))
    assert bench.measure_traffic_live(L, B, 0) is None


def test_sharded_value_is_refused_unless_every_rank_reports_a_communicator_of_the_whole_job():
    """bench.py --gpus N prints `mul_relin_sharded.value` under the claim "one RCCL all-gather inside the library" only when every rank's
    ncclCommCount (and the library's own view) equals N (cuhe/CuHE.cu:217-256 is the multi-GPU path it stands for)."""
    import bench

    def info(rank, cnt, lib_ranks=None, urank=None, state="initialised"):
        return ("rank %d: comm_init ok; rccl 22606; communicator %s; ncclCommCount %d, ncclCommUserRank %d (library: %d ranks, rank %d); exchanges so far 0 "
                "(ncclAllGather in place 0, padded 0, broadcast group 0), last: none") % (rank, state, cnt, rank if urank is None else urank,
                                                                                         cnt if lib_ranks is None else lib_ranks, rank)
    for world in (2, 4, 8):
        good = [info(r, world) for r in range(world)]
        assert bench.sharded_comm_guard(good, world, True) is None
        assert bench.sharded_comm_guard(good, world, False) is None                     # torch.distributed exchange: nothing claimed
        one_rank_comms = [info(r, 1) for r in range(world)]                              # every rank alone in its own communicator
        assert "ncclCommCount 1" in bench.sharded_comm_guard(one_rank_comms, world, True)
        short = good[:-1] + [info(world - 1, world - 1)]
        assert "rank %d" % (world - 1) in bench.sharded_comm_guard(short, world, True)
        assert bench.sharded_comm_guard(good[:-1] + [None], world, True) is not None     # a rank's report is missing
        assert bench.sharded_comm_guard(good[:-1], world, True) is not None
        assert bench.sharded_comm_guard([info(0, world)] + [info(r, world, urank=0) for r in range(1, world)], world, True) is not None
        assert bench.sharded_comm_guard([info(r, world, state="not initialised") for r in range(world)], world, True) is not None
        assert bench.sharded_comm_guard([info(r, world, lib_ranks=1) for r in range(world)], world, True) is not None
    assert bench.sharded_comm_guard(["RCCL unavailable: librccl.so not found"] * 2, 2, True) is not None


def test_gpu_local_cpus_from_a_sysfs_tree(tmp_path, monkeypatch):
    """bench_aux/placement.py: the CPUs local to the n-th AMD GPU function in bus order (what bench.py narrows itself to before any HIP call;
    the library's cuhe_hip_pin_thread_to_device reads the same files), visible-device lists followed, anything unparsable -> unknown."""
    from bench_aux.placement import gpu_local_cpus
    def dev(name, vendor, cls, cpus):
        d = tmp_path / name; d.mkdir()
        (d / "vendor").write_text(vendor + "\n"); (d / "class").write_text(cls + "\n"); (d / "local_cpulist").write_text(cpus + "\n")
    dev("0000:85:00.0", "0x1002", "0x120000", "64-127,192-255")
    dev("0000:05:00.0", "0x1002", "0x120000", "0-63,128-191")
    dev("0000:01:00.0", "0x8086", "0x020000", "0-63")                # a NIC
    dev("0000:03:00.0", "0x1002", "0x060400", "0-63")                # an AMD bridge: not a GPU
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    text, cpus = gpu_local_cpus(0, str(tmp_path))
    assert text == "0-63,128-191" and cpus == set(range(0, 64)) | set(range(128, 192))
    text, cpus = gpu_local_cpus(1, str(tmp_path))
    assert text == "64-127,192-255" and 200 in cpus and 5 not in cpus
    assert gpu_local_cpus(2, str(tmp_path)) == ("", set())
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,0")
    assert gpu_local_cpus(0, str(tmp_path))[0] == "64-127,192-255"
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-deadbeef")
    assert gpu_local_cpus(0, str(tmp_path)) == ("", set())


def test_bench_process_pins_itself_and_a_multi_gpu_child_gets_everything_back():
    """bench_aux/placement.py in a process of its own (the affinity of the pytest process is left alone): pin_process_to_gpu narrows the process to the
    GPU's local CPUs it may use and records it; restore_original_affinity (the preexec hook of the N > 1 PRINCE child, whose threads place
    themselves per device) gives everything back."""
    import subprocess
    import sys
    code = r'''
import os, sys
sys.path.insert(0, %r)
from bench_aux import placement
allowed = sorted(os.sched_getaffinity(0))
if len(allowed) < 2:
    print("SKIP"); sys.exit(0)
local = set(allowed[: len(allowed) // 2])
placement.gpu_local_cpus = lambda index, base=None: (",".join(map(str, sorted(local))), set(local) | {100000})
rec = placement.pin_process_to_gpu(0)
assert rec["pinned"] and rec["cpus_allowed"] == len(local) and rec["of"] == len(allowed), rec
assert os.sched_getaffinity(0) == local
placement.restore_original_affinity()
assert os.sched_getaffinity(0) == set(allowed)
placement.gpu_local_cpus = lambda index, base=None: ("", set())
assert placement.pin_process_to_gpu(0)["pinned"] is False and os.sched_getaffinity(0) == set(allowed)
print("OK")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() in ("OK", "SKIP"), r.stdout + r.stderr
