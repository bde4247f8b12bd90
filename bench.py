#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X cuHE backend.

Metric (BASELINE.json): 64K-point forward NTT/s per node (u32[32768] zero-padded
input -> u64[65536] natural-order output over P = 2^64-2^32+1: the contract of
the reference's ntt_{1,2,3}_64k kernels that doc/Perf_NTT.txt times), with the
HBM roofline fraction of that transform and, as further figures in the same
JSON line, DHS ciphertext mul+relin/s (BASELINE config 4), the full multiply
(config 3) and the homomorphic PRINCE block (config 5).

A "step" = one pass of the hot path over one batch of `--batch` independent
64K-point transforms, inputs already resident in HBM.  One process per GPU
(torch.distributed / RCCL only for the barrier + max-over-ranks timing: the
transforms are independent, so the path shards with no data-path collective).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --perf-table profiles/rNN_perf_ntt_table.txt      # the table of doc/Perf_NTT.txt on this GPU
"""
import argparse
import ctypes as C
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC for RCCL (already exported on the GPU boxes)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
RING_PARAMS = {"2^15": (25, 2, 16, 576, 24, 65536),       # x^32768 + 1: 48 primes < 2^24, 72 keys (the reference's largest ring)
               "2^16": (25, 2, 16, 552, 23, 131072)}      # x^65536 + 1: BASELINE config 4 read literally, 48 primes < 2^23, 69 keys


def code_only(text):
    """a source text without comments and blank space: what the hash below is taken over, so that editing a comment does not
    orphan a measurement"""
    import re
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    lines = []
    for line in text.split("\n"):
        line = re.sub(r"\s+", " ", re.sub(r"//.*$", "", line)).strip()
        if line:
            lines.append(line)
    return "\n".join(lines)


def kernel_sha16():
    """identifies the transform kernels a committed PMC figure was measured on (code of the three kernel headers, comments removed)"""
    h = hashlib.sha256()
    for f in ("modp.cuh", "ntt_kernels.cuh", "ntt_onewg.cuh"):
        h.update(code_only(open(os.path.join(ROOT, "cuhe_amd", "csrc", f)).read()).encode())
    return h.hexdigest()[:16]


class SmiSampler:
    """Shader clock and socket power of one GPU sampled from a background thread while a load runs (the
    limiter of the transforms -- integer issue under the chip's power limit -- belongs in the driver-run record, not in a
    builder-side microbenchmark).  Sources, in order: the amdsmi python binding of the ROCm image, then the amdgpu sysfs files;
    neither present -> every figure is None and `source` says so.  Nothing here touches the GPU's queues."""

    def __init__(self, index):
        import threading
        self.index, self.source, self.samples, self._stop, self._thr = index, None, [], threading.Event(), None
        self._read = None
        try:
            sys.path.append("/opt/rocm/share/amd_smi")
            import amdsmi
            amdsmi.amdsmi_init()
            h = amdsmi.amdsmi_get_processor_handles()[index]

            def read():
                clk = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                pw = amdsmi.amdsmi_get_power_info(h)
                w = pw.get("current_socket_power", pw.get("average_socket_power", pw.get("socket_power")))
                return (float(clk.get("clk", clk.get("cur_clk"))), float(w) if isinstance(w, (int, float)) else None)
            read()
            self._read, self.source = read, "amdsmi"
        except Exception as ex:
            self._err = repr(ex)[:120]
        if self._read is None:
            try:
                cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
                base = os.path.dirname(cards[index])
                hw = (glob.glob(os.path.join(base, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(base, "hwmon", "hwmon*", "power1_input")) + [None])[0]

                def read():
                    mhz = None
                    for line in open(os.path.join(base, "pp_dpm_sclk")):
                        if "*" in line:
                            mhz = float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
                    return (mhz, float(open(hw).read()) / 1e6 if hw else None)
                read()
                self._read, self.source = read, "sysfs"
            except Exception as ex:
                self._err = getattr(self, "_err", "") + " | " + repr(ex)[:120]

    def _loop(self, period):
        while not self._stop.is_set():
            try:
                self.samples.append(self._read())
            except Exception:
                pass
            self._stop.wait(period)

    def start(self, period=0.01):
        import threading
        self.samples = []
        self._stop.clear()
        if self._read:
            self._thr = threading.Thread(target=self._loop, args=(period,), daemon=True)
            self._thr.start()

    def stop(self):
        self._stop.set()
        if self._thr:
            self._thr.join()
            self._thr = None
        return self.summary(self.samples)

    def idle(self):
        try:
            return self.summary([self._read()]) if self._read else None
        except Exception:
            return None

    @staticmethod
    def summary(samples):
        def stat(vals):
            vals = [v for v in vals if v is not None]
            return {"mean": round(sum(vals) / len(vals), 1), "min": round(min(vals), 1), "max": round(max(vals), 1)} if vals else None
        return {"samples": len(samples), "sclk_mhz": stat([c for c, _ in samples]), "socket_power_w": stat([w for _, w in samples])}


def measure_traffic_live(L, B, chunk):
    """HBM-side traffic and VALU lane-instructions of the headline kernel(s), measured IN THIS RUN: three child runs of this script
    (--pmc-child: 3 launches of the same batch) under `rocprofv3 --kernel-trace --pmc <one counter>` -- separate passes, no other
    trace domain, as MI355X_MICROARCH.md prescribes -- read back from the rocpd databases.  Units and corrections as in
    tools/make_traffic_json.py: FETCH_SIZE / WRITE_SIZE are KB per dispatch, FETCH_SIZE counts wide streaming reads at half their
    bytes on gfx950 (x 2), SQ_INSTS_VALU is wave-instructions summed per shader engine (32 samples per dispatch, x 64 lanes).
    Returns None when rocprofv3 is not there or a pass fails (the committed record is used then)."""
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    per = {}
    tmp = tempfile.mkdtemp(prefix="cuhe_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
                   "--batch", str(B), "--len", str(L), "--chunk", str(chunk), "--no-cpu", "--no-mulrelin", "--no-prince", "--no-limiter", "--no-pmc"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=240)
            dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None
            db = sqlite3.connect(dbs[0])
            rows = db.execute("select s.display_name, count(*), avg(e.value) from rocpd_pmc_event e join rocpd_info_pmc i on e.pmc_id = i.id"
                              " join rocpd_kernel_dispatch d on d.event_id = e.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id"
                              " where i.name = ? group by s.display_name", (counter,)).fetchall()
            for name, n, avg in rows:
                for key in ("ntt_onewg_stream<15, 0, 0>", "ntt_pass1w<16, 0>", "ntt_pass2w<16, 0>"):
                    if key in name:
                        per.setdefault(key, {})[counter] = (int(n), float(avg))
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    ow, p1, p2 = "ntt_onewg_stream<15, 0, 0>", "ntt_pass1w<16, 0>", "ntt_pass2w<16, 0>"
    full = lambda k: k in per and len(per[k]) == 3
    if full(ow):
        c = per[ow]
        return {"one_launch": True, "kernel": ow, "dispatches_sampled": c["FETCH_SIZE"][0], "transforms_per_launch": B,
                "bytes_per_launch": int(1024.0 * (2 * c["FETCH_SIZE"][1] + c["WRITE_SIZE"][1])),
                "fetch_bytes_per_launch_x2": int(2048.0 * c["FETCH_SIZE"][1]), "write_bytes_per_launch": int(1024.0 * c["WRITE_SIZE"][1]),
                "valu_lane_instructions_per_transform": int(c["SQ_INSTS_VALU"][1] * 32 * 64 / B)}
    if full(p1) and full(p2):
        per_pair = min(B, chunk if chunk else (256 << 20) // (L * 8))
        b = 1024.0 * (2 * (per[p1]["FETCH_SIZE"][1] + per[p2]["FETCH_SIZE"][1]) + per[p1]["WRITE_SIZE"][1] + per[p2]["WRITE_SIZE"][1])
        return {"one_launch": False, "kernel": p1 + " + " + p2, "dispatches_sampled": per[p1]["FETCH_SIZE"][0], "transforms_per_launch": per_pair, "bytes_per_launch": int(b),
                "valu_lane_instructions_per_transform": int((per[p1]["SQ_INSTS_VALU"][1] + per[p2]["SQ_INSTS_VALU"][1]) * 32 * 64 / per_pair)}
    return None


def measure_limiter(lib, ck, torch, step, local_rank, seconds=1.0):
    """what the chip does under (a) the timed step itself and (b) a dense stream of the 64-bit integer instructions the field
    arithmetic lowers to, both for about `seconds`: shader clock and socket power from the SMI, the dense stream's sustained
    rate and clock from the library's probe kernel (cuhe_hip_probe_valu, s_memtime of its own waves)."""
    out = {"source": None}
    smi = SmiSampler(local_rank)
    out["source"] = smi.source or ("unavailable: " + getattr(smi, "_err", "no amdsmi, no sysfs"))
    torch.cuda.synchronize()
    time.sleep(0.25)
    out["idle"] = smi.idle()
    # (a) the benchmarked step, back to back
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); one = max(time.perf_counter() - t0, 1e-4)
    n = max(3, int(seconds / one))
    smi.start()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    rec = smi.stop()
    rec.update({"steps": n, "seconds": round(wall, 3), "ms_per_step": round(wall / n * 1e3, 4)})
    out["under_the_timed_step"] = rec
    # (b) the dense instruction stream at the occupancies the transforms run at
    dense = {}
    for wps in (2, 4):
        r, mhz, cyc = C.c_double(0), C.c_double(0), C.c_double(0)
        smi.start()
        ck(lib.cuhe_hip_probe_valu(0, wps, int(seconds * 500), C.byref(r), C.byref(mhz), C.byref(cyc)))
        rec = smi.stop()
        rec.update({"lane_instructions_T_per_s": round(r.value / 1e12, 2), "shader_mhz_from_s_memtime": round(mhz.value, 1), "cycles_per_wave_instruction_per_simd": round(cyc.value, 3)})
        dense["%d_waves_per_simd" % wps] = rec
    out["dense_64bit_integer_stream"] = dense
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # default: ~1 s of GPU time in the timed region (300 steps x ~3.3 ms), so that samplers outside the process see it
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=5)
    # 8192 transforms (4 GiB of output) per step: a step lasts ~4 ms, so that even a 5-step timed region is long enough
    # for the clocks to settle (the chip needs tens of ms of load; with 1024 per step a 10-step run reads 20 % low)
    ap.add_argument("--batch", type=int, default=8192, help="64K-point transforms per step per GPU")
    ap.add_argument("--len", type=int, default=65536, dest="length")
    ap.add_argument("--chunk", type=int, default=0, help="transforms per launch pair (0 = library default)")
    ap.add_argument("--overlap", type=int, default=0, help="1: pass-1/pass-2 two-stream pipeline, 0: serial launches (default)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (single-GPU smoke test of the N>1 path)")
    ap.add_argument("--relin-batch", type=int, default=32, help="ciphertexts per call of the batched multiply+relinearise leg")
    ap.add_argument("--relin-lanes", type=int, default=0, help="streams the batched multiply+relinearise spreads groups of 4 ciphertexts over (0 = library default)")
    ap.add_argument("--relin-threads", type=int, default=4, help="host threads of the concurrent multiply+relinearise leg (4 ciphertexts per call each)")
    ap.add_argument("--mul-batch", type=int, default=16, help="operand pairs per call of the batched full-multiply leg")
    ap.add_argument("--cyclic", action="store_true", help="ciphertext legs: keep the reference's cyclic 2n-point representation on x^n+1 rings (A/B against the negacyclic default)")
    ap.add_argument("--no-mulrelin", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="transforms in the CPU baseline sample (0 = auto)")
    ap.add_argument("--ring", choices=list(RING_PARAMS), default="2^16",
                    help="ciphertext mul+relin leg: x^32768+1 (the reference's largest ring, 64K-point cyclic or 32K-point negacyclic "
                         "transforms) or x^65536+1 (BASELINE config 4 read literally: 64K-point negacyclic transforms, 23-bit primes); "
                         "the other ring is reported beside it unless --one-ring")
    ap.add_argument("--one-ring", action="store_true")
    ap.add_argument("--no-prince", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes (roofline.traffic then comes from the committed profiles/traffic_r*.json)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)       # the process the live PMC passes profile: 1 + 2 steps of the headline batch
    ap.add_argument("--no-limiter", action="store_true", help="skip the ~3 s clock / power / dense-stream measurement (roofline.valu_ceiling.live)")
    ap.add_argument("--perf-table", default="", help="write the bundle-size table of doc/Perf_NTT.txt (tests/test_ntt.cu:140-151) to this file and exit")
    args = ap.parse_args()
    args.relin_params = RING_PARAMS[args.ring]

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    single_dev = os.environ.get("CUHE_BENCH_SINGLE_DEVICE") == "1"
    # a stuck rank prints every thread's stack and exits (the launcher then stops the others): on request, and by default
    # after 15 minutes in a multi-rank run (a collective that never completes would otherwise hold the node until the caller's limit)
    watchdog = int(os.environ.get("CUHE_BENCH_WATCHDOG", "900" if world > 1 else "0"))
    if watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(watchdog, exit=True)
    if single_dev:
        local_rank = 0                      # test hook: every rank on device 0 (use with --dist-backend gloo)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend)
    assert world == args.gpus or world == 1, "launch with --nproc-per-node == --gpus"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from cuhe_amd import capi                      # raises if the HIP library is missing
    lib, ck = capi.lib, capi.check
    ck(lib.cuhe_hip_set_device_base(local_rank))
    if args.chunk:
        ck(lib.cuhe_hip_set_ntt_chunk(args.chunk))
    ck(lib.cuhe_hip_set_ntt_overlap(args.overlap))
    if args.relin_lanes:
        ck(lib.cuhe_hip_set_relin_lanes(args.relin_lanes))

    if args.perf_table:
        perf_table(lib, ck, torch, dev, args.perf_table)
        return

    L, B = args.length, args.batch
    # synthetic input: uniform 32-bit words, generated on the device
    gen = torch.Generator(device=dev); gen.manual_seed(0xC0FFEE + rank)
    src = torch.randint(-(1 << 31), (1 << 31) - 1, (B, L // 2), dtype=torch.int32, device=dev, generator=gen)
    dst = torch.empty((B, L), dtype=torch.int64, device=dev)
    ck(lib.cuhe_hip_ntt_prepare(L, 0))

    def step():
        ck(lib.cuhe_hip_ntt_fwd_batched(dst.data_ptr(), src.data_ptr(), L, B, L // 2, 0, None))

    if args.pmc_child:               # profiled by measure_traffic_live(): nothing but the headline kernel, three launches
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        return

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ntt_per_s = world * B * args.steps / dt

    # ---- second figure at N > 1: one ciphertext multiply+relinearise with its CRT primes sharded over the ranks
    sharded = None
    if world > 1 and not args.no_mulrelin:
        try:
            sharded = bench_mulrelin_sharded(lib, ck, torch, np, dist, dev, rank, world, args, single_dev)
        except Exception as ex:                      # never lose the headline line to the secondary figure
            sharded = {"error": repr(ex)[:300]}
        ck(lib.cuhe_hip_ntt_prepare(L, 0))

    # ---- third figure at N > 1: every rank multiplies + relinearises its OWN ciphertexts (keys replicated, no exchange):
    # the throughput form of multi-GPU use, next to the latency form above
    replicated = None
    if world > 1 and not args.no_mulrelin:
        barrier()
        mr, err = None, None
        try:
            mr = bench_mulrelin(lib, ck, torch, np, dev, args, args.relin_params)
            vals = [mr["value"], (mr["batched"] or {}).get("value", 0.0), (mr["concurrent"] or {}).get("value", 0.0), 1.0]
        except Exception as ex:                      # every rank still joins the reduction below
            err, vals = repr(ex)[:300], [0.0, 0.0, 0.0, 0.0]
        v = torch.tensor(vals, dtype=torch.float64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        replicated = {"unit": "mul+relin/s, all GPUs (independent ciphertexts per GPU, keys replicated)", "one_at_a_time": round(float(v[0]), 1),
                      "batched": round(float(v[1]), 1), "concurrent": round(float(v[2]), 1), "ranks_reporting": int(v[3]), "rank0": mr if err is None else {"error": err}}
        ck(lib.cuhe_hip_ntt_prepare(L, 0))

    # ---- homomorphic PRINCE (BASELINE config 5) across the N GPUs of this node: rank 0 runs the in-process multi-device
    # driver (one host thread per GPU, examples/Prince/Prince.cu:194-200) while the other ranks wait
    prince = None
    if not args.no_prince:
        if world > 1:
            barrier()
        if rank == 0:
            prince = bench_prince(world, single_dev)
        if world > 1:
            barrier()

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel pair, timed live with hipEvents on the launch stream
        ms1, ms2, mst = C.c_float(0), C.c_float(0), C.c_float(0)
        iters = max(3, min(args.steps, 10))
        ck(lib.cuhe_hip_time_ntt_fwd(dst.data_ptr(), src.data_ptr(), L, B, iters, 0, None,
                                     C.byref(ms1), C.byref(ms2), C.byref(mst)))
        n_tr = iters * B
        alg_bytes = 10 * L                              # 4*(L/2) + 8*L per transform
        # the transform is the launch pair pass1+pass2; in production the pair is software-pipelined over chunks
        # (pass 2 of chunk c overlaps pass 1 of chunk c+1), so the pair's duration is taken from hipEvents that
        # bracket the pipelined region on the launch stream; the serial per-pass durations are reported beside it.
        pair_s = mst.value * 1e-3
        achieved = n_tr * alg_bytes / pair_s / 1e9
        # which kernels ran: the persistent one-workgroup transform is ONE launch for the whole batch (its events read ~0 for the
        # first span); the two-pass pair works through the batch in launch pairs of one 256 MiB slab each
        one_launch = ms1.value < 0.02 * mst.value
        per_launch = B if one_launch else min(B, args.chunk if args.chunk else (256 << 20) // (L * 8))
        # PMC-derived figures (HBM-side bytes, VALU lane-instructions per transform) cannot be collected inside the timed
        # process.  They are MEASURED IN THIS RUN by child processes under rocprofv3 --pmc (measure_traffic_live, below); if that
        # is not possible (no rocprofv3, N > 1, --no-pmc) they come from the committed passes of this same command
        # (profiles/traffic_r*.json), and only if that file was measured on the kernels that are running now (hash of the kernel
        # sources); otherwise null.
        rec = {}
        traffic, lane_instr, pmc_note = None, None, "no profiles/traffic_r*.json for this transform length"
        try:
            tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")))
            if tf and L == 65536:
                rec = json.load(open(tf[-1]))
                if rec.get("kernel_sha16") == kernel_sha16() and bool(rec.get("one_launch")) == one_launch:
                    traffic = rec["bytes_per_transform"] * per_launch
                    lane_instr = rec.get("valu_lane_instructions_per_transform")
                    pmc_note = ("from committed rocprofv3 --pmc passes of this command (%s), NOT measured in this run: HBM-side bytes per launch (%d transforms); "
                                "algorithmic = %d" % (os.path.basename(tf[-1]), per_launch, per_launch * alg_bytes))
                else:
                    pmc_note = "%s was measured on other kernel sources (%s, now %s): re-run tools/profile_final.sh" % (os.path.basename(tf[-1]), rec.get("kernel_sha16"), kernel_sha16())
        except Exception as ex:
            pmc_note = "traffic file unreadable: %r" % (ex,)
        committed = {"traffic": traffic, "note": pmc_note}
        live = None
        if world == 1 and L == 65536 and not args.no_pmc:
            try:
                live = measure_traffic_live(L, B, args.chunk)
            except Exception:
                live = None
        if live and bool(live["one_launch"]) == one_launch and live["transforms_per_launch"] == per_launch:
            traffic, lane_instr = live["bytes_per_launch"], live["valu_lane_instructions_per_transform"]
            rec = dict(rec)
            rec.setdefault("dense_stream_ceiling_T_per_s", 36.5)           # (profiles/r02_valu_cost_model.txt; re-measured below as valu_ceiling.live)
            pmc_note = ("MEASURED IN THIS RUN: rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU: one counter per pass) of a child process that launches "
                        "the same batch three times; HBM-side bytes per launch = 2 x FETCH_SIZE (gfx950 counts wide streaming reads at half their bytes) + WRITE_SIZE = %d + %d; "
                        "algorithmic = %d" % (live.get("fetch_bytes_per_launch_x2", 0), live.get("write_bytes_per_launch", 0), per_launch * alg_bytes))
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "dispatch": dispatch_info(lib),
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": pmc_note,
                    "kernel": ("ntt_onewg_stream<kOutU64> (persistent one-workgroup transform: two 32K-point halves per row, one launch per call)" if one_launch
                               else "ntt_pass1w<16,0> + ntt_pass2w<16,0> (one transform = one launch pair)"), "kernel_sha16": kernel_sha16(),
                    "algorithmic_bytes_per_transform": alg_bytes,
                    "pipelined_ms_per_batch": round(mst.value / iters, 4),
                    "pass1_ms_per_batch": round(ms1.value / iters, 4), "pass2_ms_per_batch": round(ms2.value / iters, 4)}
        if live:
            roofline["traffic_live"] = live
            roofline["traffic_committed_record"] = committed
        limiter = None
        if not args.no_limiter:
            try:
                limiter = measure_limiter(lib, ck, torch, step, local_rank)
            except Exception as ex:
                limiter = {"error": repr(ex)[:300]}
        if lane_instr:
            # the limiter that actually binds: integer VALU issue under the chip's power limit.  Dense
            # streams of these instructions saturate near 36.5 T lane-instructions/s (profiles/r02_valu_cost_model.txt)
            got = n_tr * lane_instr / pair_s / 1e12
            ceil_t = rec.get("dense_stream_ceiling_T_per_s")        # measured figure, carried by the same hash-guarded record
            roofline["valu_ceiling"] = {"lane_instructions_per_transform": lane_instr, "achieved_T_per_s": round(got, 2),
                                        "dense_stream_ceiling_T_per_s": ceil_t, "frac_of_dense_stream_ceiling": round(got / ceil_t, 3) if ceil_t else None,
                                        "note": "ceiling = pure v_lshl_add_u64 / v_mad_u64_u32 / v_cmp_u64 streams at 4 waves per SIMD (profiles/r02_valu_cost_model.txt)"}
        if limiter is not None:
            # measured IN THIS RUN: clock and power under the timed step and under the dense stream, the dense stream's own rate
            vc = roofline.setdefault("valu_ceiling", {})
            vc["live"] = limiter
            d4 = (limiter.get("dense_64bit_integer_stream") or {}).get("4_waves_per_simd") or {}
            if lane_instr and d4.get("lane_instructions_T_per_s"):
                vc["frac_of_live_dense_stream"] = round(n_tr * lane_instr / pair_s / 1e12 / d4["lane_instructions_T_per_s"], 3)

        # ---- the shorter zero-padded transforms of the reference contract (cuhe/Base.cu:309-437, 492-608): same bytes per step
        # as the 64K-point batch.  Their sub-transforms of 16K / 8K points run in the one-workgroup form (ntt_onewg.cuh)
        other_lengths = {}
        if L == 65536:
            for L2 in (32768, 16384):
                try:
                    B2 = B * (L // L2)
                    ck(lib.cuhe_hip_ntt_prepare(L2, 0))
                    s2 = src.view(-1)[:B2 * (L2 // 2)].view(B2, L2 // 2)
                    d2 = dst.view(-1)[:B2 * L2].view(B2, L2)
                    a1, a2, at = C.c_float(0), C.c_float(0), C.c_float(0)
                    ck(lib.cuhe_hip_time_ntt_fwd(d2.data_ptr(), s2.data_ptr(), L2, B2, 3, 0, None, C.byref(a1), C.byref(a2), C.byref(at)))
                    per = at.value * 1e-3 / (3 * B2)
                    other_lengths[str(L2)] = {"value": round(1.0 / per, 1), "unit": "NTT/s", "batch": B2, "frac": round(10 * L2 / per / 1e9 / HBM_PEAK_GBS, 4),
                                              "kernel": "ntt_onewg<%d, zero-padded halves> (one launch)" % (13 if L2 == 16384 else 14), "dispatch": dispatch_info(lib)}
                except Exception as ex:
                    other_lengths[str(L2)] = {"error": repr(ex)[:200]}
        roofline["other_lengths"] = other_lengths
        step(); torch.cuda.synchronize()                     # (the shorter transforms borrowed dst: the 64K-point outputs are checked below)

        # measured device-to-device copy ceiling of this box: 1 GiB read + 1 GiB written per copy
        ca = torch.empty(1 << 28, dtype=torch.int32, device=dev); cb = torch.empty_like(ca)
        cb.copy_(ca); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            cb.copy_(ca)
        e1.record(); torch.cuda.synchronize()
        roofline["measured_copy_GBs"] = round(5 * 2 * ca.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del ca, cb

        # ---- correctness spot check against the oracle (never timed, never shipped)
        cpu = None
        if not args.no_cpu:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            # rows of the first round, of the middle of the persistent walk and the last one (both halves of a rendezvous pair)
            checked_rows = sorted({0, 1, B // 2, B // 2 + 1, B - 2, B - 1} & set(range(B)))
            for b in checked_rows:
                xs = src[b].cpu().numpy().view(np.uint32)
                got = dst[b].cpu().numpy().view(np.uint64)
                assert np.array_equal(got, O.ntt_ext(xs, L)), "GPU NTT differs from the oracle (row %d of %d)" % (b, B)
            roofline["checked_rows_vs_oracle"] = checked_rows
        if not args.no_cpu and world == 1:              # rank 0 at N = 1 only (torchrun pins OMP_NUM_THREADS=1 per rank)
            # ---- CPU baseline: the oracle's O(L log L) transform on the host cores, bounded sample
            cores = os.cpu_count() or 1
            sample = args.cpu_sample or max(cores * 24, 64)          # ~3 s wall on all host cores
            xh = np.random.default_rng(1).integers(0, 1 << 32, (sample, L // 2), dtype=np.uint32)
            O.ntt_ext_batch(xh[:cores], L, 0)            # warm-up
            t1 = time.perf_counter()
            _, used = O.ntt_ext_batch(xh, L, 0)
            cdt = time.perf_counter() - t1
            cpu = {"value": round(sample / cdt, 1), "unit": "NTT/s", "cores": used, "kind": "port",
                   "sample": "%d 64K-point forward transforms, oracle radix-2 NTT (u128 %% P), OpenMP over transforms, %.1f s"
                             % (sample, cdt)}

        mulrelin = mulrelin2 = mulfull = None
        if not args.no_mulrelin and world == 1:
            mulrelin = bench_mulrelin(lib, ck, torch, np, dev, args, args.relin_params)
            if not args.one_ring and not args.cyclic:
                other = [k for k in RING_PARAMS if k != args.ring][0]
                try:
                    mulrelin2 = bench_mulrelin(lib, ck, torch, np, dev, args, RING_PARAMS[other])
                except Exception as ex:
                    mulrelin2 = {"error": repr(ex)[:300]}
            mulfull = bench_mul_full(lib, ck, torch, np, dev, with_cpu=not args.no_cpu, batch=args.mul_batch, cyclic=args.cyclic)

        out = {
            "metric": "64K-point fwd NTT/s (u32[32768] -> u64[65536] over P=2^64-2^32+1)",
            "value": round(ntt_per_s, 1), "unit": "NTT/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            # the reference's best published figure for this exact metric (doc/Perf_NTT.txt:14,
            # bundle 512: 0.0226647 ms per 64K transform = 44 121 NTT/s on one unstated NVIDIA GPU)
            "higher_is_better": True, "scaling": "weak", "vs_baseline": round(ntt_per_s / 44121.0, 2),
            "dtype": "u64 (mod 2^64-2^32+1)", "data": "synthetic",
            "config": {"workload": "batched 64K-point forward NTT, %d transforms per step per GPU, reference contract "
                                   "of ntt_{1,2,3}_64k (cuhe/Base.cu:659-785)" % B,
                       "transform_len": L, "batch_per_gpu": B, "sharding": "independent transforms per rank, no collective"},
            "roofline": roofline, "cpu_baseline": cpu,
            "reference_best_published": {"value": 44121, "unit": "NTT/s", "hardware": "unstated NVIDIA GPU",
                                         "source": "doc/Perf_NTT.txt:14 (bundle 512)"},
            "mul_relin_single_frac_hbm": {(r or {}).get("params", {}).get("transform", "?"): (r or {}).get("frac_hbm") for r in (mulrelin, mulrelin2) if r},
            "mul_relin": mulrelin, "mul_relin_other_ring": mulrelin2, "mul_relin_sharded": sharded, "mul_relin_replicated": replicated,
            "mul_full": mulfull, "prince": prince,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def perf_table(lib, ck, torch, dev, path):
    """doc/Perf_NTT.txt on this GPU: ms per single forward transform when `bundle` transforms share a launch pair, bundle
    1 ... 512, lengths 16K / 32K / 64K, 1024 transforms per measurement on consecutive slabs (tests/test_ntt.cu:67-100,140-151)."""
    cnt = 1024
    ref = {1: (0.0486284, 0.051598, 0.064822), 512: (0.00407564, 0.00804859, 0.0226647)}     # doc/Perf_NTT.txt:5,14
    rows = []
    lens = (16384, 32768, 65536)
    for L in lens:
        ck(lib.cuhe_hip_ntt_prepare(L, 0))
    src = {L: torch.randint(-(1 << 31), (1 << 31) - 1, (cnt, L // 2), dtype=torch.int32, device=dev) for L in lens}
    dst = {L: torch.empty((cnt, L), dtype=torch.int64, device=dev) for L in lens}
    for bundle in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
        row = [bundle]
        for L in lens:
            s, d = src[L], dst[L]

            def run():
                for b0 in range(0, cnt, bundle):
                    ck(lib.cuhe_hip_ntt_fwd_batched(d[b0:].data_ptr(), s[b0:].data_ptr(), L, bundle, L // 2, 0, None))
            run(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); run(); torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / cnt * 1e3)
            row.append(best)
        rows.append(row)
    with open(path, "w") as f:
        f.write("# forward NTT (u32 half-length input -> u64 output), ms per single transform with `Num` transforms per launch pair;\n")
        f.write("# 1024 transforms per measurement on consecutive slabs, host launch loop + device time, best of 3 (bench.py --perf-table);\n")
        f.write("# the shape of the reference's doc/Perf_NTT.txt (tests/test_ntt.cu:140-151; its hardware is not stated):\n")
        f.write("#   reference bundle 1:   16K %.7f  32K %.7f  64K %.7f\n#   reference bundle 512: 16K %.7f  32K %.7f  64K %.7f\n" % (ref[1] + ref[512]))
        f.write("%-6s %-14s %-14s %-14s\n" % ("Num", "16K", "32K", "64K"))
        for r in rows:
            f.write("%-6d %-14.7f %-14.7f %-14.7f\n" % tuple(r))
    print(open(path).read())


def dispatch_info(lib):
    """which kernel form the last transform call of this thread took, and the rendezvous give-up count (cuhe_hip_last_dispatch_info)"""
    buf = C.create_string_buffer(256)
    return buf.value.decode() if lib.cuhe_hip_last_dispatch_info(0, buf, 256) == 0 else None


def bench_prince(world, single_dev):
    """BASELINE config 5: wall clock of one homomorphic PRINCE block (examples/Prince/Prince.cu:83-87 times princeEncrypt)
    and its known answer (Prince.cu:96), gates on arrays of ciphertexts, S-boxes of a layer spread over `world` GPUs by
    the in-process multi-device driver (tests/cxx/test_prince_arrays_cxx.cpp)."""
    try:
        exe = os.path.join(ROOT, "cuhe_amd", "lib", "test_prince_arrays_cxx")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cuhe_amd", "cxx"), "-s", exe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        # three blocks in one process: `value` is the FIRST (what the reference times: one block after set-up, examples/Prince/Prince.cu:83-87);
        # the later ones no longer pay the first-time hipMalloc of the arrays (profiles/r05_prince_gaps_arrays.txt)
        # (one GPU only: the repeated form of the multi-device client has not run on hardware)
        cmd = [exe, "--no-round-checks", "--async", "--json"] + (["--repeat", "3"] if world == 1 else []) + ["--devices", str(world)] + (["--virtual"] if single_dev and world > 1 else [])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": (r.stdout[-300:] + r.stderr[-300:]).strip()}
        recs = [json.loads(l) for l in line]
        rec = recs[0]
        out = {"value": rec["prince_seconds"], "unit": "s per PRINCE block (64 ciphertext bits, 1920 cAnd, 1152 relin, 24 levels)", "n_gpus": rec["devices"],
               "virtual_devices": rec["virtual"], "known_answer": rec["kat"], "known_answer_ok": rec["kat_ok"],
               "params": "CuDHS(25,2,16,25,25,21845): n=16384, 32K-point transforms, 25 -> 1 primes, 40 keys",
               "client": "CuCtxtArray (not the reference's call pattern)",
               "mode": "CuCtxtArray gates, asynchronous, one host thread per GPU (Prince.cu:194-200)",
               "later_blocks_same_process": [x["prince_seconds"] for x in recs[1:]],
               "all_known_answers_ok": all(x["kat_ok"] for x in recs)}
        out["gate_by_gate"] = bench_prince_gate_by_gate()
        return out
    except Exception as ex:
        return {"error": repr(ex)[:300]}


def bench_prince_gate_by_gate():
    """The reference client's call pattern (examples/Prince/Prince.cu:204-322: ONE host thread, the default stream, one
    CuCtxt gate per call) on one GPU, same block and known answer: with the reference's synchronise-per-gate semantics
    (cuhe/CuHE.cu:98,121,139,157) and with the library's scheduled gates (CuHE.h setScheduled / CUHE_SCHED=1: the same
    client code, independent gates issued concurrently by the library's worker threads)."""
    try:
        exe = os.path.join(ROOT, "cuhe_amd", "lib", "test_prince_flow")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cuhe_amd", "cxx"), "-s", exe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        # one synchronous block, then four scheduled blocks in the same process: the first is the figure of rounds 4 (workers and their
        # scratch are new), the later ones are what a client that encrypts more than one block sees
        r = subprocess.run([exe, "--threads", "1", "--no-round-checks", "--compare", "--repeat", "4"], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, CUHE_SCHED_STATS="1"))
        secs, blocks = {}, []
        for l in r.stdout.splitlines():
            if l.startswith("Prince Encryption:"):
                if "scheduled gates" in l:
                    blocks.append(float(l.split()[2]))
                else:
                    secs["sync_1thread"] = float(l.split()[2])
            elif l.startswith("batches:"):
                secs["scheduler"] = l.strip()[:200]
        if blocks:
            # the first scheduled block of a process is the warm-up (new worker threads grow their scratch: several GB of first-time hipMalloc,
            # 0.08-0.17 s from run to run); the figure is the median of the three blocks after it, like every timed region of this script
            later = sorted(blocks[1:])
            secs["scheduled_1thread"] = later[len(later) // 2] if later else blocks[0]
            secs["scheduled_1thread_first_block"] = blocks[0]
            secs["scheduled_1thread_blocks"] = blocks
        ok = r.returncode == 0 and r.stdout.count("9fb51935fc3df524   expected 9fb51935fc3df524   right") == 5 and len(blocks) == 4 and "sync_1thread" in secs
        if not ok:
            return {"error": (r.stdout[-300:] + r.stderr[-300:]).strip()}
        secs.update({"unit": "s per PRINCE block, CuCtxt gates one per call from one host thread (the reference client's pattern)", "known_answer_ok": True})
        return secs
    except Exception as ex:
        return {"error": repr(ex)[:300]}


def sharded_comm_guard(per_rank, world, in_library):
    """SURVEY 8(e) / cuhe/CuHE.cu:217-256: a `mul_relin_sharded` value quoted as "exchange inside the library over RCCL" is only printed
    when EVERY rank's communicator really spans the job: ncclCommCount == world, ncclCommUserRank == the rank, and the library's own
    view agrees (cuhe_hip_comm_info, gathered from all ranks).  Returns None when the record may be printed, else the reason.  (When the
    in-library communicator is not in use -- the torch.distributed exchange around the same stages -- there is nothing to check.)"""
    import re
    if not in_library:
        return None
    if len(per_rank) != world or any(not isinstance(x, str) for x in per_rank):
        return "communicator reports of %d rank(s) for a job of %d" % (sum(isinstance(x, str) for x in per_rank), world)
    for r, text in enumerate(per_rank):
        m = re.search(r"ncclCommCount (-?\d+), ncclCommUserRank (-?\d+) \(library: (-?\d+) ranks, rank (-?\d+)\)", text)
        if not m or "communicator initialised" not in text:
            return "rank %d: no initialised communicator in its report (%s)" % (r, text[:120])
        cnt, urank, lranks, lrank = (int(v) for v in m.groups())
        if cnt != world or lranks != world:
            return "rank %d: ncclCommCount %d / library %d ranks in a job of %d" % (r, cnt, lranks, world)
        if urank != r or lrank != r:
            return "rank %d reports ncclCommUserRank %d / library rank %d" % (r, urank, lrank)
    return None


def bench_mulrelin_sharded(lib, ck, torch, np, dist, dev, rank, world, args, single_dev=False):
    """primes of ONE ciphertext sharded over the ranks, one all-gather (CRT rows before ICRT) per
    multiply+relinearise; value = multiplies per second of the whole job (max time over ranks).  The whole chain, RCCL
    all-gather included, is one C-ABI call per multiply (cuhe_hip_mul_relin_sharded) enqueued on the compute stream;
    if the in-library communicator cannot be made (e.g. every rank on one device in the gloo smoke test) the exchange
    falls back to torch.distributed around the same stage functions (cuhe_amd/sharded.py)."""
    from cuhe_amd import capi
    from cuhe_amd.sharded import HipBackend, ShardedMulRelin
    d, p, w, mn, cut, m = args.relin_params
    # local set-up first; the ranks then agree (one all-reduce) that everybody is ready before the first data-path
    # collective, so that a local failure (e.g. out of memory) cannot leave the others hanging in the all-gather
    ready, err = 1, None
    try:
        lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters()
        ck(lib.cuhe_hip_set_parameters(d, p, w, mn, cut, m))
        ck(lib.cuhe_hip_init(None, 0))
        q = capi.get_params()
        K, W = q.numEvalKey, lib.cuhe_hip_words_coeff(0)
        ek = np.random.default_rng(7).integers(0, 1 << 32, (K, q.rawLen, W), dtype=np.uint32)
        ek[:, :, W - 1] &= 0x7FFF
        # each rank keeps the evaluation keys of ITS primes only (keys partitioned with the primes)
        kf, kc = C.c_int(), C.c_int()
        ck(lib.cuhe_hip_key_range(world, rank, C.byref(kf), C.byref(kc)))
        ck(lib.cuhe_hip_init_relin_range(ek.ctypes.data_as(C.c_void_p), kf.value, kc.value))
        lib_ct_len = lib.cuhe_hip_ct_len()
        key_bytes = kc.value * K * lib_ct_len * 8
        hb = HipBackend()
        sh = ShardedMulRelin(hb, 0, rank, world)
        gen = torch.Generator(device=dev); gen.manual_seed(5)
        a = torch.randint(0, 1 << (q.logCrtPrime - 1), (q.numCrtPrime, q.crtLen), dtype=torch.int32, device=dev, generator=gen)
        b = torch.randint(0, 1 << (q.logCrtPrime - 1), (q.numCrtPrime, q.crtLen), dtype=torch.int32, device=dev, generator=gen)
        na = hb.ntt_rows(sh.own(a).contiguous()); nb = hb.ntt_rows(sh.own(b).contiguous())
        outc = torch.zeros((sh.count, q.crtLen), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
    except Exception as ex:
        ready, err = 0, repr(ex)[:200]
    flag = torch.tensor([ready], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters()
        return {"error": "set-up failed on at least one rank" + (": " + err if err else "")}
    # in-library communicator: rank 0 makes the id, torch.distributed carries the 128 bytes
    # (byte 128 carries rank 0's verdict: if it could not make the id NOBODY calls comm_init -- the others would block in it)
    idt = torch.zeros(129, dtype=torch.uint8, device=dev)
    if rank == 0:
        uid = (C.c_uint8 * 128)()
        # every rank on ONE device (the gloo smoke test): RCCL refuses duplicate devices, and a communicator set-up that one rank
        # has left while the other still waits in it has no time-out -- it is not attempted at all
        made = (not single_dev) and lib.cuhe_hip_comm_unique_id(uid) == 0
        idt = torch.tensor(list(uid) + [1 if made else 0], dtype=torch.uint8, device=dev)
    dist.broadcast(idt, 0)
    host_id = [int(v) for v in idt.cpu().tolist()]
    ok = host_id[128]
    if ok:
        uid = (C.c_uint8 * 128)(*host_id[:128])
        ok = 1 if lib.cuhe_hip_comm_init(world, rank, uid) == 0 else 0
    comm_err = None if ok else ("every rank on one device" if single_dev else lib.cuhe_hip_last_error().decode()[:200])
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    in_library = int(flag.item()) == 1
    # what RCCL itself says on every rank (version, ncclCommCount, ncclCommUserRank) next to the library's view: gathered to
    # every rank so that rank 0 can print it -- the first thing to read if the first N > 1 contact misbehaves
    cbuf = C.create_string_buffer(512)
    lib.cuhe_hip_comm_info(cbuf, 512)
    my_info = "rank %d: comm_init %s; %s" % (rank, "ok" if ok else "FAILED (%s)" % comm_err, cbuf.value.decode())
    per_rank = [None] * world
    try:
        dist.all_gather_object(per_rank, my_info)
    except Exception as ex:
        per_rank = [my_info, "all_gather_object failed: %r" % (ex,)]
    if not in_library:
        lib.cuhe_hip_comm_destroy()
        if comm_err is None:
            comm_err = "comm_init failed on another rank"

    def one():
        if in_library:
            ck(lib.cuhe_hip_mul_relin_sharded(outc.data_ptr(), na.data_ptr(), nb.data_ptr(), 0, 0, None))
            return outc
        return sh.mul_relin(na, nb)
    if in_library:
        # the first call through RCCL is allowed to fail (the exact failing call and RCCL's message arrive in the error
        # string): every rank then falls back to the torch.distributed exchange together, and the leg still delivers
        lib_err = None
        try:
            first = one().clone()
            torch.cuda.synchronize()
        except Exception as ex:
            lib_err = repr(ex)[:300]
        flag = torch.tensor([0 if lib_err else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            in_library = False
            comm_err = "first cuhe_hip_mul_relin_sharded failed: " + (lib_err or "on another rank")
            lib.cuhe_hip_comm_destroy()
    if in_library:                                     # same rows through the torch.distributed exchange: must agree
        assert torch.equal(first, sh.mul_relin(na, nb)), "in-library all-gather differs from the torch.distributed path"
        lib.cuhe_hip_comm_info(cbuf, 512)
        per_rank[rank] = per_rank[rank] + " | after the first call: " + cbuf.value.decode()
    else:
        first = one().clone()
    for _ in range(3):
        one()
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    # the part every participant repeats (ICRT of the gathered rows, window extraction, the k window transforms): timed as
    # ICRT + the key switch onto ONE prime; its share of the call is the serial fraction that bounds the speed-up
    rows_all = torch.randint(0, 1 << (q.logCrtPrime - 1), (q.numCrtPrime, q.crtLen), dtype=torch.int32, device=dev, generator=gen)
    raw = torch.zeros((q.rawLen, W), dtype=torch.int32, device=dev)
    acc1 = torch.empty((1, lib.cuhe_hip_ct_len()), dtype=torch.int64, device=dev)
    def replicated():
        ck(lib.cuhe_hip_icrt(raw.data_ptr(), rows_all.data_ptr(), lib.cuhe_hip_log_coeff(0), 0, None))
        ck(lib.cuhe_hip_relin_range(acc1.data_ptr(), raw.data_ptr(), 0, sh.first, 1, 0, None))
    for _ in range(3):
        replicated()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        replicated()
    torch.cuda.synchronize()
    t_rep = (time.perf_counter() - t0) / reps
    lib_comm_size = lib.cuhe_hip_comm_size() if in_library else None
    lib.cuhe_hip_comm_destroy()
    lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters()
    refused = sharded_comm_guard(per_rank, world, in_library)
    if refused:                                         # no number under a claim that did not hold
        return {"value": None, "error": "not reported: " + refused, "rccl_per_rank": per_rank, "comm_size": lib_comm_size}
    return {"value": round(1.0 / dt, 2), "unit": "mul+relin/s (one ciphertext, primes sharded)", "ms": round(dt * 1e3, 3),
            "key_bytes_per_rank": key_bytes, "key_primes_per_rank": kc.value,
            "replicated_ms": round(t_rep * 1e3, 3), "serial_fraction": round(t_rep / dt, 3),
            "serial_note": "ICRT + window extraction + the k window transforms are repeated on every rank (exchanging the transformed windows instead "
                           "would move k*n*8 = %d B per multiply against %d B of CRT rows); the key-switch inner product, both inverse transforms and the "
                           "key memory divide by the number of ranks" % (K * lib_ct_len * 8, q.numCrtPrime * q.crtLen * 4),
            "primes_per_rank": sh.count, "numCrtPrime": q.numCrtPrime, "numEvalKey": K, "ring_degree": q.modLen,
            "comm_size": lib_comm_size, "rccl_per_rank": per_rank,
            "exchange": "RCCL all-gather inside cuhe_hip_mul_relin_sharded, on the compute stream (one in-place ncclAllGather when the blocks are equal, one padded ncclAllGather otherwise; the path every rank took is in rccl_per_rank)" if in_library
                        else "torch.distributed all-gather around the C-ABI stages (in-library communicator unavailable: %s)" % comm_err,
            "collective": "1 all-gather of %d B per rank per multiply" % (sh.count * q.crtLen * 4)}


def bench_mul_full(lib, ck, torch, np, dev, with_cpu=True, batch=16, cyclic=False):
    """BASELINE config 3: N = 2^15 (64K-point transforms), 32 CRT primes, full multiply of two raw polynomials
    CRT -> NTT -> pointwise -> INTT (+ reduction mod x^n+1) -> ICRT, device resident (mulZZX without the ZZX<->raw staging)."""
    from cuhe_amd import capi
    d, p, w, mn, cut, m = 9, 2, 16, 576, 24, 65536
    lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters()
    ck(lib.cuhe_hip_set_negacyclic(0 if cyclic else -1))
    ck(lib.cuhe_hip_set_parameters(d, p, w, mn, cut, m))
    ck(lib.cuhe_hip_init(None, 0))
    q = capi.get_params()
    npn, L, W, logq = q.numCrtPrime, lib.cuhe_hip_ct_len(), lib.cuhe_hip_words_coeff(0), lib.cuhe_hip_log_coeff(0)
    rep = "negacyclic %d-point" % L if lib.cuhe_hip_ct_negacyclic() else "cyclic %d-point" % L
    gen = torch.Generator(device=dev); gen.manual_seed(9)
    ra = torch.randint(-(1 << 31), (1 << 31) - 1, (q.rawLen, W), dtype=torch.int32, device=dev, generator=gen)
    rb = torch.randint(-(1 << 31), (1 << 31) - 1, (q.rawLen, W), dtype=torch.int32, device=dev, generator=gen)
    ca = torch.zeros((npn, q.crtLen), dtype=torch.int32, device=dev); cb = torch.zeros_like(ca)
    na = torch.empty((npn, L), dtype=torch.int64, device=dev); nb = torch.empty_like(na)
    out = torch.zeros((q.rawLen, W), dtype=torch.int32, device=dev)

    def one():
        ck(lib.cuhe_hip_crt(ca.data_ptr(), ra.data_ptr(), logq, 0, None))
        ck(lib.cuhe_hip_crt(cb.data_ptr(), rb.data_ptr(), logq, 0, None))
        ck(lib.cuhe_hip_ct_ntt(na.data_ptr(), ca.data_ptr(), logq, 0, None))
        ck(lib.cuhe_hip_ct_ntt(nb.data_ptr(), cb.data_ptr(), logq, 0, None))
        ck(lib.cuhe_hip_ct_mul(na.data_ptr(), na.data_ptr(), nb.data_ptr(), logq, 0, None))
        ck(lib.cuhe_hip_ct_intt(ca.data_ptr(), na.data_ptr(), logq, 1, 0, None))
        ck(lib.cuhe_hip_icrt(out.data_ptr(), ca.data_ptr(), logq, 0, None))

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    got = out.cpu().numpy().view(np.uint32)
    ha, hb = ra.cpu().numpy().view(np.uint32), rb.cpu().numpy().view(np.uint32)
    # ---- the same multiplication for B independent operand pairs per call (cuhe_hip_mul_raw_batch)
    batched = None
    try:
        B = batch
        # B DISTINCT operand pairs; every result row is compared with the single chain on the same pair
        rab = torch.randint(-(1 << 31), (1 << 31) - 1, (B * q.rawLen, W), dtype=torch.int32, device=dev, generator=gen)
        rbb = torch.randint(-(1 << 31), (1 << 31) - 1, (B * q.rawLen, W), dtype=torch.int32, device=dev, generator=gen)
        outb = torch.empty((B * q.rawLen, W), dtype=torch.int32, device=dev)
        for _ in range(2):
            ck(lib.cuhe_hip_mul_raw_batch(outb.data_ptr(), rab.data_ptr(), rbb.data_ptr(), 0, B, 0, None))
        torch.cuda.synchronize()
        keep_a, keep_b = ra.clone(), rb.clone()
        for i in range(B):
            ra.copy_(rab[i * q.rawLen:(i + 1) * q.rawLen]); rb.copy_(rbb[i * q.rawLen:(i + 1) * q.rawLen])
            one()
            assert torch.equal(outb[i * q.rawLen:(i + 1) * q.rawLen], out), "batched result %d differs from the single chain" % i
        ra.copy_(keep_a); rb.copy_(keep_b); one(); torch.cuda.synchronize()
        breps = max(3, 64 // B)
        t0 = time.perf_counter()
        for _ in range(breps):
            ck(lib.cuhe_hip_mul_raw_batch(outb.data_ptr(), rab.data_ptr(), rbb.data_ptr(), 0, B, 0, None))
        torch.cuda.synchronize()
        bdt = (time.perf_counter() - t0) / breps / B
        batched = {"value": round(1.0 / bdt, 1), "unit": "full multiplies/s (raw -> raw)", "ms_per_multiply": round(bdt * 1e3, 4), "batch": B,
                   "checked": "%d distinct operand pairs, every result equal to the single chain" % B}
    except Exception as ex:
        batched = {"error": repr(ex)[:300]}
    lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters(); lib.cuhe_hip_set_negacyclic(-1)
    res = {"value": round(1.0 / dt, 1), "unit": "full multiplies/s (raw -> raw)", "ms": round(dt * 1e3, 4),
           "params": {"setParameters": [d, p, w, mn, cut, m], "numCrtPrime": npn, "transform": rep, "coeff_words": W},
           "transforms_per_multiply": 3 * npn, "batched": batched}
    if with_cpu:
        # the same multiply on the host through the oracle with OpenMP over the CRT primes on all cores (checker + reported
        # CPU baseline, never the product path), and -- when the box has libgmp -- the way the reference's host library does
        # it: ONE big-integer multiplication of Kronecker-packed operands (NTL, which the reference calls at
        # examples/DHS/DHS.cu:219-221, is not installed in this image; it builds on GMP)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        o = O.Ctx(d, p, w, mn, cut, m)
        q0 = o.coeff_modulus(0)
        used = O.set_threads(0)
        o.mul_raw(ha, hb, 0)                              # warm-up (thread pool, page faults)
        t1 = time.perf_counter()
        want = o.mul_raw(ha, hb, 0)
        cdt = time.perf_counter() - t1
        O.set_threads(1)
        o.close()
        assert np.array_equal(got, want), "GPU full multiply differs from the oracle"
        res["cpu_baseline"] = {"value": round(1.0 / cdt, 3), "unit": "full multiplies/s", "cores": used, "kind": "port",
                               "sample": "1 multiply (N=2^15, 32 primes) through oracle/oracle.c, OpenMP over the CRT primes, %.2f s" % cdt}
        t1 = time.perf_counter()
        gm = O.gmp_mul_xn1(ha, hb, q0)
        gdt = time.perf_counter() - t1
        res["cpu_baseline_gmp"] = None
        if gm is not None:
            assert np.array_equal(gm, want), "GMP product differs from the oracle"
            res["cpu_baseline_gmp"] = {"value": round(1.0 / gdt, 3), "unit": "full multiplies/s", "cores": 1, "kind": "port",
                                       "sample": "1 multiply: Kronecker substitution + one mpz_mul + coefficient reduction (libgmp opened at run time; "
                                                 "stand-in for NTL's ZZX multiply, which is not installed), %.2f s" % gdt}
    return res


def bench_mulrelin(lib, ck, torch, np, dev, args, params):
    """DHS ciphertext multiply + relinearise per second on 64K-point transforms (BASELINE config 4 shape:
    48 CRT primes < 2^24, w = 16).  NTT-domain operands -> reduced CRT-domain result, keys resident in HBM."""
    d, p, w, mn, cut, m = params
    lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters()
    ck(lib.cuhe_hip_set_negacyclic(0 if args.cyclic else -1))
    ck(lib.cuhe_hip_set_parameters(d, p, w, mn, cut, m))
    ck(lib.cuhe_hip_init(None, 0))
    from cuhe_amd import capi
    q = capi.get_params()
    npn, L, K, W = q.numCrtPrime, lib.cuhe_hip_ct_len(), q.numEvalKey, lib.cuhe_hip_words_coeff(0)
    rep = "negacyclic %d-point" % L if lib.cuhe_hip_ct_negacyclic() else "cyclic %d-point" % L
    rng = np.random.default_rng(7)
    ek = rng.integers(0, 1 << 32, (K, q.rawLen, W), dtype=np.uint32)
    ek[:, :, W - 1] &= 0x7FFF                        # keep below 2^(32W-17): any value works, crt reduces
    t0 = time.perf_counter()
    ck(lib.cuhe_hip_init_relin(ek.ctypes.data_as(C.c_void_p)))
    init_s = time.perf_counter() - t0
    logq = lib.cuhe_hip_log_coeff(0)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    a = torch.randint(0, 1 << (q.logCrtPrime - 1), (npn, q.crtLen), dtype=torch.int32, device=dev, generator=gen)
    b = torch.randint(0, 1 << (q.logCrtPrime - 1), (npn, q.crtLen), dtype=torch.int32, device=dev, generator=gen)
    na = torch.empty((npn, L), dtype=torch.int64, device=dev); nb = torch.empty_like(na); nc = torch.empty_like(na)
    cr = torch.empty((npn, q.crtLen), dtype=torch.int32, device=dev)
    raw = torch.zeros((q.rawLen, W), dtype=torch.int32, device=dev)
    ck(lib.cuhe_hip_ct_ntt(na.data_ptr(), a.data_ptr(), logq, 0, None))
    ck(lib.cuhe_hip_ct_ntt(nb.data_ptr(), b.data_ptr(), logq, 0, None))

    fused = True          # what CuCtxt::relin does since round 5; the two-call form is timed beside it below

    def one():
        ck(lib.cuhe_hip_ct_mul(nc.data_ptr(), na.data_ptr(), nb.data_ptr(), logq, 0, None))       # cAnd
        ck(lib.cuhe_hip_ct_intt(cr.data_ptr(), nc.data_ptr(), logq, 1, 0, None))                   # relin: x2r
        ck(lib.cuhe_hip_icrt(raw.data_ptr(), cr.data_ptr(), logq, 0, None))
        if fused:       # relinearization ; n2c as the one call CuCtxt::relin makes (round 5)
            ck(lib.cuhe_hip_relin_crt(cr.data_ptr(), raw.data_ptr(), 0, 0, None))
        else:
            ck(lib.cuhe_hip_relinearization(nc.data_ptr(), raw.data_ptr(), 0, 0, None))
            ck(lib.cuhe_hip_ct_intt(cr.data_ptr(), nc.data_ptr(), logq, 1, 0, None))                   # n2c

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    single_dispatch = dispatch_info(lib)              # (of the chain's last transform call: the inverse rows of the result)
    # the same chain with relinearization and n2c as two calls (rounds 1-4): same results
    single_variants = {}
    try:
        ref = cr.clone()
        fused = False
        one(); torch.cuda.synchronize()
        assert torch.equal(cr, ref), "single chain (two calls) differs"
        t0 = time.perf_counter()
        for _ in range(reps):
            one()
        torch.cuda.synchronize()
        single_variants["two_calls_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
    except Exception as ex:
        single_variants["error"] = repr(ex)[:200]
    fused = True
    key_bytes = 8 * K * npn * L
    # ---- the same chain for B independent ciphertexts per call (cuhe_hip_mul_relin_batch): every stage runs over
    # B*np rows and a key value fetched from HBM serves four ciphertexts; results are bit-identical (checked below)
    batched = None
    try:
        B = args.relin_batch
        # B DISTINCT ciphertext pairs; every result row is compared with the single chain on the same pair
        ab = torch.randint(0, 1 << (q.logCrtPrime - 1), (B * npn, q.crtLen), dtype=torch.int32, device=dev, generator=gen)
        bb = torch.randint(0, 1 << (q.logCrtPrime - 1), (B * npn, q.crtLen), dtype=torch.int32, device=dev, generator=gen)
        nab = torch.empty((B * npn, L), dtype=torch.int64, device=dev); nbb = torch.empty_like(nab)
        for i in range(B):
            ck(lib.cuhe_hip_ct_ntt(nab[i * npn:].data_ptr(), ab[i * npn:].data_ptr(), logq, 0, None))
            ck(lib.cuhe_hip_ct_ntt(nbb[i * npn:].data_ptr(), bb[i * npn:].data_ptr(), logq, 0, None))
        out = torch.empty((B * npn, q.crtLen), dtype=torch.int32, device=dev)
        for _ in range(2):
            ck(lib.cuhe_hip_mul_relin_batch(out.data_ptr(), nab.data_ptr(), nbb.data_ptr(), 0, B, 0, None))
        torch.cuda.synchronize()
        keep_a, keep_b = na.clone(), nb.clone()
        singles = []
        for i in range(B):
            na.copy_(nab[i * npn:(i + 1) * npn]); nb.copy_(nbb[i * npn:(i + 1) * npn])
            one()
            singles.append(cr.clone())
            assert torch.equal(out[i * npn:(i + 1) * npn], cr), "batched result %d differs from the single chain" % i
        na.copy_(keep_a); nb.copy_(keep_b); one(); torch.cuda.synchronize()
        breps = max(6, 40 // B)
        t0 = time.perf_counter()
        for _ in range(breps):
            ck(lib.cuhe_hip_mul_relin_batch(out.data_ptr(), nab.data_ptr(), nbb.data_ptr(), 0, B, 0, None))
        torch.cuda.synchronize()
        bdt = (time.perf_counter() - t0) / breps / B
        # algorithmic minimum per ciphertext of a batch: the keys once per call, two ct-domain operands in, one CRT result out
        alg = key_bytes / B + 2 * 8 * npn * L + 4 * npn * q.modLen
        batched = {"value": round(1.0 / bdt, 2), "unit": "mul+relin/s", "ms_per_ciphertext": round(bdt * 1e3, 4), "batch": B,
                   "key_bytes_per_ciphertext": key_bytes // min(B, 16),
                   "algorithmic_bytes_per_ciphertext": int(alg), "frac_hbm": round(alg / bdt / 1e9 / HBM_PEAK_GBS, 4),
                   "checked": "%d distinct ciphertext pairs, every result equal to the single chain" % B,
                   "note": "B independent chains per call on one stream; products formed on load by the inverse transforms; inverse CRT column sums on the matrix cores (int8 MFMA, base-128 digits of the residue products); key-switch inner product on the matrix cores (int8 MFMA over signed base-256 digits) in tiles of 16 ciphertexts"}
        # twice the batch (the keys are amortised over more ciphertexts): the same operands twice, the two halves of the result equal
        try:
            B2 = 2 * B
            nab2, nbb2 = torch.cat((nab, nab)), torch.cat((nbb, nbb))
            out2 = torch.empty((B2 * npn, q.crtLen), dtype=torch.int32, device=dev)
            for _ in range(2):
                ck(lib.cuhe_hip_mul_relin_batch(out2.data_ptr(), nab2.data_ptr(), nbb2.data_ptr(), 0, B2, 0, None))
            torch.cuda.synchronize()
            assert torch.equal(out2[:B * npn], out) and torch.equal(out2[B * npn:], out), "batch of %d differs from the batch of %d" % (B2, B)
            t0 = time.perf_counter()
            for _ in range(max(4, breps // 2)):
                ck(lib.cuhe_hip_mul_relin_batch(out2.data_ptr(), nab2.data_ptr(), nbb2.data_ptr(), 0, B2, 0, None))
            torch.cuda.synchronize()
            b2dt = (time.perf_counter() - t0) / max(4, breps // 2) / B2
            batched["twice_the_batch"] = {"batch": B2, "ms_per_ciphertext": round(b2dt * 1e3, 4), "value": round(1.0 / b2dt, 2), "checked": "both halves equal the batch of %d" % B}
            del nab2, nbb2, out2
        except Exception as ex:
            batched["twice_the_batch"] = {"error": repr(ex)[:200]}
    except Exception as ex:
        batched = {"error": repr(ex)[:300]}
    # ---- the batched call from several host threads at once (own stream and own scratch each: the library is
    # re-entrant): the HBM-bound inner product of one call overlaps the instruction-bound transforms of another
    concurrent = None
    try:
        import threading
        T, Bc, creps = args.relin_threads, 4, 10
        bufs = []
        for _ in range(T):
            st = C.c_void_p(); ck(lib.cuhe_hip_stream_create(0, C.byref(st)))
            lo = (len(bufs) * Bc) % max(1, B - Bc + 1)                      # a different slice of the distinct pairs per thread
            bufs.append((st, nab[lo * npn:(lo + Bc) * npn].contiguous(), nbb[lo * npn:(lo + Bc) * npn].contiguous(),
                         torch.empty((Bc * npn, q.crtLen), dtype=torch.int32, device=dev), lo))
        torch.cuda.synchronize()

        def work(t, n):
            st, x, y, o, _ = bufs[t]
            for _ in range(n):
                ck(lib.cuhe_hip_mul_relin_batch(o.data_ptr(), x.data_ptr(), y.data_ptr(), 0, Bc, 0, st))
            ck(lib.cuhe_hip_stream_sync(0, st))
        for t in range(T):
            work(t, 1)
        th = [threading.Thread(target=work, args=(t, creps)) for t in range(T)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        cdt = (time.perf_counter() - t0) / (T * Bc * creps)
        for bf in bufs:
            for i in range(Bc):
                assert torch.equal(bf[3][i * npn:(i + 1) * npn], singles[bf[4] + i]), "concurrent result differs from the single chain"
        for bf in bufs:
            ck(lib.cuhe_hip_stream_destroy(0, bf[0]))
        concurrent = {"value": round(1.0 / cdt, 2), "unit": "mul+relin/s", "ms_per_ciphertext": round(cdt * 1e3, 4), "host_threads": T, "batch": Bc,
                      "note": "T host threads, one stream each, batched calls of 4 ciphertexts"}
    except Exception as ex:
        concurrent = {"error": repr(ex)[:300]}
    lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters(); lib.cuhe_hip_set_negacyclic(-1)
    return {"value": round(1.0 / dt, 2), "unit": "mul+relin/s", "ms": round(dt * 1e3, 3),
            "params": {"setParameters": [d, p, w, mn, cut, m], "ring_degree": q.modLen, "numCrtPrime": npn, "numEvalKey": K, "transform": rep},
            "algorithmic_bytes": key_bytes, "achieved_GBs": round(key_bytes / dt / 1e9, 1),
            "frac_hbm": round(key_bytes / dt / 1e9 / HBM_PEAK_GBS, 4), "key_upload_s": round(init_s, 2), "dispatch_last_transform": single_dispatch,
            "chain": "ct_mul ; ct_intt ; icrt ; relin_crt (relinearization + n2c as one call)", "variants": single_variants,
            "batched": batched, "concurrent": concurrent}


if __name__ == "__main__":
    main()
