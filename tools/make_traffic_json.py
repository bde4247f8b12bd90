#!/usr/bin/env python3
"""Turns the PMC passes of tools/profile_final.sh (pmc_FETCH_SIZE.txt, pmc_WRITE_SIZE.txt, pmc_SQ_INSTS_VALU.txt: the
rocpd_summary lines of the two 64K forward-transform kernels) into the record bench.py reads as profiles/traffic_rNN.json:
HBM-side bytes per transform (FETCH_SIZE x 2 for wide streaming reads on gfx950, MI355X_MICROARCH.md; WRITE_SIZE as is;
both in KB per dispatch), VALU lane-instructions per transform (SQ_INSTS_VALU is summed per shader engine: 32 samples per
dispatch, x 64 lanes), and the hash of the kernel sources they were measured on.
usage: make_traffic_json.py <dir with pmc_*.txt> <round tag>"""
import hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, tag = sys.argv[1], sys.argv[2]
PER_LAUNCH, L = 512, 65536            # bench.py default: 256 MiB slab = 512 transforms of 64K points per launch pair


def pmc(counter):
    """{kernel: (n, avg)} of the forward-transform kernels of the timed loop"""
    res = {}
    for line in open(os.path.join(d, "pmc_%s.txt" % counter)):
        m = re.match(r"void cuhe::(ntt_pass[12]w<16, 0>|ntt_onewg_stream<15, 0, 0>).*?%s\s+n=(\d+)\s+avg=\s*([0-9.]+)" % counter, line)
        if m:
            res[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return res


sys.path.insert(0, ROOT)
import bench                                   # the hash bench.py checks: code of the kernel headers, comments removed
fetch, write, valu = pmc("FETCH_SIZE"), pmc("WRITE_SIZE"), pmc("SQ_INSTS_VALU")
common = {"source": "profiles/%s_ntt64k_pmc.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU, separate passes with --kernel-trace only; "
                    "FETCH_SIZE x2: gfx950 counts wide streaming reads at half their bytes)" % tag,
          "command": "python bench.py --steps 2 --warmup 1 --no-mulrelin --no-cpu --no-prince",
          "kernel_sha16": bench.kernel_sha16(), "transform_len": L,
          # issue rate of dense streams of the instructions the field arithmetic lowers to (tools/ubench_rates.hip, 4 waves per SIMD)
          "dense_stream_ceiling_T_per_s": 36.5, "dense_stream_ceiling_source": "profiles/r02_valu_cost_model.txt"}
ow = "ntt_onewg_stream<15, 0, 0>"        # <32K-point halves, row source = zero-padded u32, output = u64>
if ow in fetch and ow in write and ow in valu:
    # the persistent one-workgroup transform: ONE launch per call of BATCH transforms (bench.py default: 8192)
    BATCH = 8192
    b = 1024.0 * (2 * fetch[ow][1] + write[ow][1])
    lanes = valu[ow][1] * 32 * 64 / (BATCH * L)              # SQ_INSTS_VALU: one sample per shader engine and dispatch
    rec = dict(common, one_launch=True, kernel=ow, transforms_per_launch=BATCH, bytes_per_launch=int(b), bytes_per_transform=int(b / BATCH),
               valu_lane_instructions_per_point={"one_workgroup": round(lanes, 2)}, valu_lane_instructions_per_transform=int(lanes * L))
else:
    p1, p2 = "ntt_pass1w<16, 0>", "ntt_pass2w<16, 0>"
    bytes_pair = 1024.0 * (2 * (fetch[p1][1] + fetch[p2][1]) + write[p1][1] + write[p2][1])
    lane = lambda k: valu[k][1] * 32 * 64 / (PER_LAUNCH * L)
    rec = dict(common, one_launch=False, transforms_per_launch_pair=PER_LAUNCH, bytes_per_launch_pair=int(bytes_pair), bytes_per_transform=int(bytes_pair / PER_LAUNCH),
               valu_lane_instructions_per_point={"pass1": round(lane(p1), 2), "pass2": round(lane(p2), 2)},
               valu_lane_instructions_per_transform=int((lane(p1) + lane(p2)) * L))
print(json.dumps(rec, indent=1))
