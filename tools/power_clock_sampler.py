#!/usr/bin/env python3
"""Sample socket power and the shader clock at >= 10 Hz while a command runs (VERDICT r5 "do this" #1a).

    python tools/power_clock_sampler.py OUT.txt -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras

Sources, each optional (whatever the box offers is recorded; a missing one is named in the header):
  * the amdsmi Python binding: amdsmi_get_power_info (socket power), amdsmi_get_clock_info(GFX), amdsmi_get_gpu_metrics_info
    (average / current gfxclk per XCD, socket power, throttle status, temperatures),
  * sysfs: /sys/class/drm/card*/device/hwmon/hwmon*/{power1_average,power1_input,freq1_input}, pp_dpm_sclk.
Writes one line per sample (t, W, MHz ...) and a summary (mean / p5 / p50 / p95 / max over the samples taken while the command's GPU
work ran -- samples above 40 % of the maximum power -- and over all samples).  No GPU work is launched from this process.
"""
from __future__ import annotations

import glob
import os
import statistics
import subprocess
import sys
import threading
import time


def _sysfs_sources():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        for hw in sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*"))):
            for f in ("power1_average", "power1_input", "freq1_input", "power1_cap"):
                p = os.path.join(hw, f)
                if os.path.exists(p):
                    out.setdefault(f, p)
        p = os.path.join(card, "pp_dpm_sclk")
        if os.path.exists(p):
            out.setdefault("pp_dpm_sclk", p)
    return out


def _read(p):
    try:
        with open(p) as f:
            return f.read().strip()
    except OSError:
        return None


class Smi:
    def __init__(self, log):
        self.h = None
        self.ok = {}
        try:
            import amdsmi
            self.m = amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self.h = hs[0] if hs else None
            log("# amdsmi: %d processor handle(s)" % len(hs))
        except Exception as e:                                     # noqa: BLE001
            log("# amdsmi unavailable: %r" % (e,))
            return
        for name, fn in (("power", self._power), ("clock", self._clock), ("metrics", self._metrics)):
            try:
                v = fn()
                self.ok[name] = True
                log("# amdsmi %s sample: %r" % (name, v))
            except Exception as e:                                 # noqa: BLE001
                log("# amdsmi %s unavailable: %r" % (name, e))

    def _power(self):
        d = self.m.amdsmi_get_power_info(self.h)
        return {k: d.get(k) for k in ("socket_power", "current_socket_power", "average_socket_power", "power_limit") if k in d}

    def _clock(self):
        d = self.m.amdsmi_get_clock_info(self.h, self.m.AmdSmiClkType.GFX)
        return {k: d.get(k) for k in ("clk", "cur_clk", "max_clk", "min_clk", "clk_locked", "clk_deep_sleep") if k in d}

    def _metrics(self):
        d = self.m.amdsmi_get_gpu_metrics_info(self.h)
        keep = ("current_socket_power", "average_socket_power", "current_gfxclk", "average_gfxclk_frequency", "current_gfxclks",
                "throttle_status", "indep_throttle_status", "temperature_hotspot", "temperature_mem", "average_gfx_activity",
                "accumulation_counter", "prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc",
                "hbm_thm_residency_acc", "gfxclk_lock_status")
        return {k: d.get(k) for k in keep if k in d}

    def sample(self):
        r = {}
        for name, fn in (("power", self._power), ("clock", self._clock), ("metrics", self._metrics)):
            if self.ok.get(name):
                try:
                    r[name] = fn()
                except Exception:                                  # noqa: BLE001
                    pass
        return r


def _num(v):
    try:
        if isinstance(v, (list, tuple)):
            xs = [float(x) for x in v if isinstance(x, (int, float)) and 0 < float(x) < 60000]
            return sum(xs) / len(xs) if xs else None
        f = float(v)
        return f if 0 <= f < 1e7 else None
    except (TypeError, ValueError):
        return None


def main():
    if "--" not in sys.argv or len(sys.argv) < 4:
        print(__doc__)
        return 2
    out_path = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    period = float(os.environ.get("HRV_SAMPLER_PERIOD", "0.05"))
    lines = []
    log = lines.append
    log("# command: %s" % " ".join(cmd))
    sysfs = _sysfs_sources()
    log("# sysfs sources: %r" % (sysfs,))
    smi = Smi(log)
    samples = []
    stop = threading.Event()

    def loop():
        t0 = time.time()
        while not stop.is_set():
            t = time.time() - t0
            s = {"t": t}
            r = smi.sample() if smi.h is not None else {}
            pw = r.get("power", {})
            mt = r.get("metrics", {})
            ck = r.get("clock", {})
            w = None
            for k in ("current_socket_power", "socket_power", "average_socket_power"):
                w = w if w is not None else _num(pw.get(k))
            for k in ("current_socket_power", "average_socket_power"):
                w = w if w is not None else _num(mt.get(k))
            if w is None:
                for f in ("power1_input", "power1_average"):
                    if f in sysfs and w is None:
                        v = _num(_read(sysfs[f]))
                        w = v / 1e6 if v is not None else None
            s["W"] = w
            mhz = None
            for k in ("current_gfxclks", "current_gfxclk", "average_gfxclk_frequency"):
                mhz = mhz if mhz is not None else _num(mt.get(k))
            for k in ("clk", "cur_clk"):
                mhz = mhz if mhz is not None else _num(ck.get(k))
            if mhz is None and "freq1_input" in sysfs:
                v = _num(_read(sysfs["freq1_input"]))
                mhz = v / 1e6 if v is not None else None
            s["MHz"] = mhz
            s["xcd_MHz"] = mt.get("current_gfxclks")
            s["throttle"] = mt.get("indep_throttle_status", mt.get("throttle_status"))
            s["hot"] = mt.get("temperature_hotspot")
            s["act"] = mt.get("average_gfx_activity")
            s["ppt_acc"] = mt.get("ppt_residency_acc")
            s["acc"] = mt.get("accumulation_counter")
            samples.append(s)
            time.sleep(period)

    th = threading.Thread(target=loop, daemon=True)
    th.start()
    t0 = time.time()
    rc = subprocess.call(cmd)
    wall = time.time() - t0
    stop.set()
    th.join(timeout=2)
    log("# command exit code %d, wall %.1f s, %d samples (%.1f Hz)" % (rc, wall, len(samples), len(samples) / max(wall, 1e-9)))

    def stats(xs, unit):
        xs = sorted(x for x in xs if x is not None)
        if not xs:
            return "no samples"
        q = lambda f: xs[min(len(xs) - 1, int(f * len(xs)))]     # noqa: E731
        return "n=%d mean %.1f p5 %.1f p50 %.1f p95 %.1f max %.1f %s" % (len(xs), statistics.fmean(xs), q(0.05), q(0.5), q(0.95), xs[-1], unit)

    ws = [s["W"] for s in samples if s["W"] is not None]
    if ws:
        thr = 0.4 * max(ws)
        busy = [s for s in samples if s["W"] is not None and s["W"] >= thr]
    else:
        busy = samples
    log("# ALL samples:  power %s | shader clock %s" % (stats([s["W"] for s in samples], "W"), stats([s["MHz"] for s in samples], "MHz")))
    log("# BUSY samples (power >= 40%% of max): power %s | shader clock %s" % (stats([s["W"] for s in busy], "W"), stats([s["MHz"] for s in busy], "MHz")))
    pp = [s["ppt_acc"] for s in samples if isinstance(s.get("ppt_acc"), (int, float))]
    ac = [s["acc"] for s in samples if isinstance(s.get("acc"), (int, float))]
    if len(pp) > 1 and len(ac) > 1 and ac[-1] > ac[0]:
        log("# power-limit (PPT) residency over the run: %.3f of the firmware's accumulation ticks (ppt_residency_acc / accumulation_counter deltas)"
            % ((pp[-1] - pp[0]) / float(ac[-1] - ac[0])))
    thr_set = sorted({str(s["throttle"]) for s in samples if s.get("throttle") is not None})
    log("# throttle status values seen: %s" % (", ".join(thr_set[:8]) or "n/a"))
    log("# t_s  W  MHz  hotspot_C  gfx_activity  per-XCD MHz")
    for s in samples:
        log("%.3f %s %s %s %s %s" % (s["t"], "%.1f" % s["W"] if s["W"] is not None else "-", "%.0f" % s["MHz"] if s["MHz"] is not None else "-",
                                     s.get("hot", "-"), s.get("act", "-"), s.get("xcd_MHz", "-")))
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(l for l in lines if l.startswith("#")))
    return rc


if __name__ == "__main__":
    sys.exit(main())
