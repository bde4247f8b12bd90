#!/usr/bin/env python3
"""GPU-box probes (tools/gpu_pass.sh probes): (1) the copy-ceiling probe, every launch shape at 256 MiB / 1 GiB / 4 GiB per side; (2) thread scaling of the oracle's throughput
transform on the box's host cores (is `cores: 128` what the sample really used?); (3) mulZZX staging, synchronous vs library default."""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from cuhe_amd import capi
lib, ck = capi.lib, capi.check
torch.zeros(1, device="cuda")
print("== copy probe (GB/s, read + written)")
for size in (1 << 28, 1 << 30, 1 << 32):
    row = []
    for v in range(lib.cuhe_hip_probe_copy_shapes()):
        g = C.c_double(0)
        ck(lib.cuhe_hip_probe_copy(0, size, v, 10, C.byref(g)))
        row.append(g.value)
    print("%5d MiB per side:" % (size >> 20), " ".join("%7.0f" % x for x in row), " best: %s" % lib.cuhe_hip_probe_copy_name(int(np.argmax(row))).decode())
a = torch.empty(1 << 28, dtype=torch.int32, device="cuda"); b = torch.empty_like(a)
b.copy_(a); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): b.copy_(a)
e1.record(); torch.cuda.synchronize()
print("torch copy_ 1 GiB per side: %.0f" % (10 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9))
for i, n in enumerate(range(lib.cuhe_hip_probe_copy_shapes())): print("   shape %d: %s" % (i, lib.cuhe_hip_probe_copy_name(n).decode()))
print("== host cores: os.cpu_count() %d, sched_getaffinity %d, cgroup cpu.max %s" % (os.cpu_count(), len(os.sched_getaffinity(0)),
      open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a"))
print(subprocess.run("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)|MHz'", shell=True, capture_output=True, text=True).stdout)
import oracle_lib as O
L = 65536
x = np.random.default_rng(1).integers(0, 1 << 32, (512, L // 2), dtype=np.uint32)
O.ntt_ext_fast_batch(x[:8], L, 1)
for th in (1, 2, 4, 8, 16, 32, 64, 128):
    n = max(8, min(512, th * 8))
    O.ntt_ext_fast_batch(x[:n], L, th)
    t0 = time.perf_counter(); O.ntt_ext_fast_batch(x[:n], L, th); dt = time.perf_counter() - t0
    print("fast transform, %3d threads: %8.1f NTT/s (%.1f per thread)" % (th, n / dt, n / dt / th), flush=True)
print("== mulZZX staging")
for env, name in (({"CUHE_SCHED": "0"}, "CUHE_SCHED=0 (synchronous gates)"), ({}, "library default (scheduled gates; mulZZX runs its gates directly)")):
    e = dict(os.environ); e.pop("CUHE_SCHED", None); e.update(env)
    r = subprocess.run([os.path.join(ROOT, "cuhe_amd", "lib", "bench_mulzzx"), "10"], capture_output=True, text=True, env=e, timeout=600)
    print("## " + name); print(r.stdout + r.stderr[-300:])
