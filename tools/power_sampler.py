#!/usr/bin/env python3
"""Sidecar telemetry sampler: socket power, per-XCD gfx clocks, temperatures and throttle residency of every AMD GPU the
driver shows, at ~25 Hz, as JSON lines -- started as its OWN process before any GPU work (it never touches HIP: amdsmi reads
the driver's metrics table; the fallback reads hwmon sysfs files).

    python3 tools/power_sampler.py OUT.jsonl [hz]        stop it with SIGTERM (tools/power_probe.py does)
"""
import glob
import json
import os
import signal
import sys
import time

out_path = sys.argv[1]
hz = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
stop = False


def _stop(*_):
    global stop
    stop = True


signal.signal(signal.SIGTERM, _stop)
signal.signal(signal.SIGINT, _stop)
KEYS = ("current_socket_power", "average_socket_power", "current_gfxclk", "current_gfxclks", "average_gfxclk_frequency", "current_uclk",
        "temperature_hotspot", "temperature_mem", "throttle_status", "indep_throttle_status", "energy_accumulator", "accumulation_counter",
        "ppt_residency_acc", "prochot_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc",
        "average_gfx_activity", "average_umc_activity", "gfx_activity_acc", "mem_activity_acc", "firmware_timestamp", "voltage_gfx")


def amdsmi_source():
    import amdsmi
    amdsmi.amdsmi_init()
    handles = amdsmi.amdsmi_get_processor_handles()
    meta = []
    for h in handles:
        d = {}
        for name, fn in (("power_cap", amdsmi.amdsmi_get_power_cap_info), ("asic", amdsmi.amdsmi_get_gpu_asic_info)):
            try:
                v = fn(h)
                d[name] = {k: (x if isinstance(x, (int, float, str)) else str(x)) for k, x in v.items()}
            except Exception as e:  # noqa: BLE001
                d[name] = f"unavailable: {e}"
        try:
            d["bdf"] = amdsmi.amdsmi_get_gpu_device_bdf(h)
        except Exception as e:  # noqa: BLE001
            d["bdf"] = f"unavailable: {e}"
        meta.append(d)

    def sample():
        rows = []
        for h in handles:
            try:
                m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                rows.append({k: m.get(k) for k in KEYS if k in m})
            except Exception as e:  # noqa: BLE001
                rows.append({"error": str(e)})
        return rows
    return "amdsmi", meta, sample


def sysfs_source():
    cards = []
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = glob.glob(os.path.join(dev, "hwmon", "hwmon*"))
        if hw and os.path.exists(os.path.join(dev, "vendor")) and open(os.path.join(dev, "vendor")).read().strip() == "0x1002":
            cards.append((dev, hw[0]))
    if not cards:
        raise RuntimeError("no amdgpu hwmon directory")

    def rd(p):
        try:
            return int(open(p).read().split()[0])
        except (OSError, ValueError, IndexError):
            return None

    def sample():
        rows = []
        for dev, hw in cards:
            pw = rd(os.path.join(hw, "power1_input")) or rd(os.path.join(hw, "power1_average"))
            rows.append({"current_socket_power": None if pw is None else pw / 1e6, "current_gfxclk": (rd(os.path.join(hw, "freq1_input")) or 0) / 1e6,
                         "temperature_hotspot": (rd(os.path.join(hw, "temp2_input")) or rd(os.path.join(hw, "temp1_input")) or 0) / 1e3,
                         "average_gfx_activity": rd(os.path.join(dev, "gpu_busy_percent"))})
        return rows
    meta = [{"sysfs": dev, "power_cap": {"power_cap": (rd(os.path.join(hw, "power1_cap")) or 0)}} for dev, hw in cards]
    return "sysfs", meta, sample


errors = []
source = None
for make in (amdsmi_source, sysfs_source):
    try:
        source = make()
        source[2]()
        break
    except Exception as e:  # noqa: BLE001 -- try the next source
        errors.append(f"{make.__name__}: {e}")
        source = None
with open(out_path, "w") as fh:
    if source is None:
        fh.write(json.dumps({"meta": "no telemetry source", "errors": errors}) + "\n")
        sys.exit(3)
    name, meta, sample = source
    fh.write(json.dumps({"meta": meta, "source": name, "hz": hz, "errors": errors, "t0": time.time()}) + "\n")
    fh.flush()
    period = 1.0 / hz
    nxt = time.time()
    n = 0
    while not stop:
        t = time.time()
        fh.write(json.dumps({"t": t, "gpus": sample()}) + "\n")
        n += 1
        if n % 50 == 0:
            fh.flush()
        nxt += period
        d = nxt - time.time()
        if d > 0:
            time.sleep(d)
        else:
            nxt = time.time()
