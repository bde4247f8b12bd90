#!/usr/bin/env python3
"""RowRebase (2 KB) is a by-value kernel argument of every ntt_onewg launch.  A/B of the default library against a build without
the argument (tools/build_variant.py libcuhe_hip_norebase.so -DCUHE_OW_NO_REBASE_ARG), each in its own process, alternating:
(1) launch-bound calls -- 32 zero-padded rows of 16K / 32K points, one-workgroup form forced, 4000 calls back to back, microseconds per call
(host + device); (2) a chip-filling call of 4096 rows of 32K points, device ms per call.   usage: rebase_arg_ab.py [child]"""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from cuhe_amd import capi
    lib, ck = capi.lib, capi.check
    dev = torch.device("cuda:0")
    out = []
    ck(lib.cuhe_hip_set_onewg(2, 1))
    for L, rows, calls in ((16384, 32, 4000), (32768, 32, 4000), (32768, 4096, 30)):
        ck(lib.cuhe_hip_ntt_prepare(L, 0))
        src = torch.randint(-(1 << 31), (1 << 31) - 1, (rows, L // 2), dtype=torch.int32, device=dev)
        dst = torch.empty((rows, L), dtype=torch.int64, device=dev)
        for _ in range(20):
            ck(lib.cuhe_hip_ntt_fwd_batched(dst.data_ptr(), src.data_ptr(), L, rows, L // 2, 0, None))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            ck(lib.cuhe_hip_ntt_fwd_batched(dst.data_ptr(), src.data_ptr(), L, rows, L // 2, 0, None))
        torch.cuda.synchronize()
        out.append("%d x %d: %.2f us/call" % (rows, L, (time.perf_counter() - t0) / calls * 1e6))
    print("; ".join(out))
    sys.exit(0)
for rep in range(3):
    for name in ("libcuhe_hip.so", "libcuhe_hip_norebase.so"):
        env = dict(os.environ, CUHE_HIP_LIB=os.path.join(ROOT, "cuhe_amd", "lib", name))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=600)
        print("%-28s %s" % (name, r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED: " + r.stderr[-300:]), flush=True)
