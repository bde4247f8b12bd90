"""HBM-side traffic and VALU lane-instructions of the headline kernel from rocprofv3 --pmc passes of a child process of bench.py."""
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import time

from .record import HBM_PEAK_GBS, ROOT

BENCH = os.path.join(ROOT, "bench.py")


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
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "p", "--", sys.executable, BENCH, "--pmc-child",
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
