"""Clock, socket power and the dense 64-bit integer stream under the timed step (roofline.valu_ceiling.live)."""
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import time

from .record import HBM_PEAK_GBS, ROOT


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
