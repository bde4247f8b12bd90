"""CPU: tools/rocpd_summary.py --between (round 5: a kernel trace cut to the timed part of a client by the marker kernel the clients launch with
CUHE_TRACE_MARK=1, with the time some kernel was running and the idle gaps of every enclosed stretch) on a small rocpd database made here."""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stretches_between_marker_kernels_and_their_idle_gaps(tmp_path):
    p = str(tmp_path / "t.db")
    db = sqlite3.connect(p)
    db.execute("create table rocpd_info_kernel_symbol(id integer, display_name text)")
    db.execute("create table rocpd_kernel_dispatch(kernel_id integer, start integer, end integer)")
    db.executemany("insert into rocpd_info_kernel_symbol values(?,?)", [(1, "cuhe::k_probe_valu(unsigned*)"), (2, "kern_a"), (3, "kern_b")])
    us = 1000
    rows = [(2, 0, 10 * us),                                            # before the first marker: not enclosed
            (1, 20 * us, 30 * us), (1, 31 * us, 40 * us),               # marker run
            (2, 50 * us, 60 * us), (3, 55 * us, 70 * us),               # two overlapping kernels (two streams)
            (2, 100 * us, 110 * us),                                    # after 30 us of idle device
            (1, 200 * us, 210 * us),                                    # marker: closes the stretch
            (3, 220 * us, 230 * us)]                                    # after the last marker: not enclosed
    db.executemany("insert into rocpd_kernel_dispatch values(?,?,?)", rows)
    db.commit(); db.close()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), "--between", "k_probe_valu", p],
                         capture_output=True, text=True, timeout=60).stdout
    assert "1 stretches enclosed" in out, out
    head = [l for l in out.splitlines() if l.startswith("-- stretch 0")][0]
    # 3 dispatches; 50 -> 110 us; some kernel running: [50, 70] + [100, 110] = 30 us; durations 10 + 15 + 10 = 35 us
    assert "3 dispatches" in head and "0.060 ms" in head and "running 0.030 ms" in head and "durations 0.035 ms" in head, head
    gaps = [l for l in out.splitlines() if "idle gaps:" in l][0]
    assert "idle gaps: 1, 0.030 ms" in gaps, gaps
    assert any("idle before kern_a" in l and "1 gaps" in l for l in out.splitlines()), out
