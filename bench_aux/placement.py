"""CPU placement of the benchmark process: the CPUs local to the GPU it drives (two-socket MI355X hosts).

A thread that launches from the far socket pays every doorbell and completion signal across the socket link, and the HIP runtime places its
host-side state (kernel-argument pools, signals, helper threads) on the node of the thread that makes the first HIP call: launch-bound legs
(a PRINCE block is ~900 small kernels) then run 10-20 % slow for the whole process (profiles/r06_numa_pinning.txt).  A benchmark harness fixes
its placement like `numactl --cpunodebind` would; the LIBRARY does the same for its own worker threads and for the thread that initialises it
(include/cuhe_hip.h: cuhe_hip_pin_thread_to_device).  Must run BEFORE torch / HIP is imported: sysfs only."""
import os


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return ""


def gpu_local_cpus(index, base="/sys/bus/pci/devices"):
    """(cpulist string, set of CPUs) of the index-th AMD GPU function in PCI bus order, honouring a plain integer list in
    ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES; ("", empty set) when unknown."""
    for name in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        e = os.environ.get(name, "")
        if not e:
            continue
        try:
            ids = [int(x) for x in e.split(",")]
        except ValueError:
            return "", set()
        if index < 0 or index >= len(ids):
            return "", set()
        index = ids[index]
    try:
        names = sorted(os.listdir(base))
    except OSError:
        return "", set()
    gpus = [n for n in names if _read(os.path.join(base, n, "vendor")) == "0x1002" and _read(os.path.join(base, n, "class"))[:4] in ("0x03", "0x12")]
    if index < 0 or index >= len(gpus):
        return "", set()
    text = _read(os.path.join(base, gpus[index], "local_cpulist"))
    cpus = set()
    for part in text.split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        try:
            cpus.update(range(int(a), int(b or a) + 1))
        except ValueError:
            return "", set()
    return text, cpus


ORIGINAL_AFFINITY = None        # what the process was allowed before pin_process_to_gpu narrowed it


def restore_original_affinity():
    """for a child that drives SEVERAL GPUs (preexec_fn): back to everything the process was allowed; its threads place themselves per device"""
    if ORIGINAL_AFFINITY:
        try:
            os.sched_setaffinity(0, ORIGINAL_AFFINITY)
        except OSError:
            pass


def pin_process_to_gpu(index):
    """Narrow this process (and what it starts afterwards) to the CPUs local to GPU `index`; returns a record for the bench line."""
    global ORIGINAL_AFFINITY
    text, cpus = gpu_local_cpus(index)
    try:
        allowed = os.sched_getaffinity(0)
    except OSError:
        return {"pinned": False, "reason": "no affinity call"}
    ORIGINAL_AFFINITY = set(allowed)
    want = allowed & cpus
    if not want or want == allowed:
        return {"pinned": False, "local_cpus": text, "reason": "unknown" if not text else "already local" if want else "no local CPU allowed"}
    os.sched_setaffinity(0, want)
    return {"pinned": True, "local_cpus": text, "cpus_allowed": len(want), "of": len(allowed)}
