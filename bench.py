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

# the auxiliary legs (bench_aux/): everything except the timed N = 1 path below
from bench_aux.record import HBM_PEAK_GBS, RING_PARAMS, code_only, kernel_sha16          # noqa: E402,F401
from bench_aux.pmc import measure_traffic_live                                            # noqa: E402,F401
from bench_aux.limiter import measure_limiter                                             # noqa: E402
from bench_aux.tables import dispatch_info, perf_table                                    # noqa: E402
from bench_aux.prince import bench_prince                                                 # noqa: E402
from bench_aux.sharded import bench_mulrelin_sharded, sharded_comm_guard                  # noqa: E402,F401
from bench_aux.ciphertext import bench_mul_full, bench_mulrelin                           # noqa: E402
from bench_aux.cpu import cpu_ntt_baseline                                                # noqa: E402
from bench_aux.placement import pin_process_to_gpu                                        # noqa: E402


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
    ap.add_argument("--no-pin", action="store_true", help="leave the process where the kernel put it (default: the CPUs local to its GPU, bench_aux/placement.py)")
    args = ap.parse_args()
    args.relin_params = RING_PARAMS[args.ring]
    # before any HIP call: the process runs next to its GPU (CUHE_BENCH_SINGLE_DEVICE: every rank drives device 0)
    placement = {"pinned": False, "reason": "--no-pin"} if args.no_pin else pin_process_to_gpu(0 if os.environ.get("CUHE_BENCH_SINGLE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0")))

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
        # `achieved` / `frac`: algorithmic bytes per launch / the kernel's launch duration from hipEvents on its launch stream (the
        # contract's definition; the committed rocprofv3 --kernel-trace --stats average of the same kernel must agree).  The same quantity
        # from the driver-visible host clock of the timed region (ms_per_step: launch, event and barrier overhead included) is printed
        # beside it as frac_from_ms_per_step; the two differ by 1-3 %.
        step_s = dt / args.steps
        achieved_step = B * alg_bytes / step_s / 1e9
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "dispatch": dispatch_info(lib),
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "achieved_basis": "hipEvent duration of the kernel on its launch stream (pipelined_ms_per_batch)",
                    "achieved_from_ms_per_step": round(achieved_step, 1), "frac_from_ms_per_step": round(achieved_step / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_note": pmc_note,
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

        # measured streaming-copy ceiling of this box: the library's float4 grid-stride copy (cuhe_hip_probe_copy),
        # 1 GiB read + 1 GiB written per launch, best of its variants; torch's copy_ (what this field held until round 5) beside it
        torch.cuda.synchronize()
        copies = {}
        for variant in range(lib.cuhe_hip_probe_copy_shapes()):
            g, name = C.c_double(0), lib.cuhe_hip_probe_copy_name(variant).decode()
            try:
                ck(lib.cuhe_hip_probe_copy(0, 1 << 30, variant, 10, C.byref(g)))
                copies[name] = round(g.value, 1)
            except Exception as ex:
                copies[name] = repr(ex)[:120]
        ca = torch.empty(1 << 28, dtype=torch.int32, device=dev); cb = torch.empty_like(ca)
        cb.copy_(ca); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            cb.copy_(ca)
        e1.record(); torch.cuda.synchronize()
        copies["torch_copy_"] = round(5 * 2 * ca.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del ca, cb
        best = [v for v in copies.values() if isinstance(v, float)]
        roofline["measured_copy_GBs"] = max(best) if best else None
        roofline["measured_copy_variants_GBs"] = copies

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
            # ---- CPU baseline: the oracle's transform on the host cores, bounded sample (bench_aux/cpu.py)
            cpu = cpu_ntt_baseline(np, L, sample=args.cpu_sample)

        mulrelin = mulrelin2 = mulfull = None
        if not args.no_mulrelin and world == 1:
            mulrelin = bench_mulrelin(lib, ck, torch, np, dev, args, args.relin_params, with_cpu=not args.no_cpu)
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
                       "transform_len": L, "batch_per_gpu": B, "sharding": "independent transforms per rank, no collective",
                       "host_placement": placement},
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


if __name__ == "__main__":
    main()
