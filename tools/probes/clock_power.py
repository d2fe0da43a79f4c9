"""Samples the GPU's shader clock and socket power from sysfs (hwmon freq1_input / power1_average|power1_input, pp_dpm_sclk) every
INTERVAL seconds while a command runs, and prints min / median / max of the samples taken while the GPU was busy (power above the
idle level seen before the command started).  usage: clock_power.py [--interval 0.02] [--label NAME] -- command ..."""
import glob
import os
import subprocess
import sys
import threading
import time


def visible_gpu_bdf():
    """PCI address of HIP device 0 of this container (the box has many cards; sysfs lists all of them)."""
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; p = torch.cuda.get_device_properties(0); "
                              "print('%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))"],
                             capture_output=True, text=True, timeout=300).stdout.strip().splitlines()
        return out[-1] if out else None
    except Exception:
        return None


def find_hwmon():
    """(device directory, hwmon directory) of the GPU this process can see, else every card that has a hwmon node."""
    bdf = visible_gpu_bdf()
    if bdf:
        dev = os.path.join("/sys/bus/pci/devices", bdf)
        hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
        if hw:
            print(f"telemetry of PCI device {bdf} (HIP device 0)")
            return [(dev, hw[0])]
        print(f"PCI device {bdf} (HIP device 0) has no hwmon node here; falling back to the first card with one")
    out = []
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for hw in sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*"))):
            out.append((card, hw))
    return out


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except Exception:
        return None


def read_dpm(path):
    try:
        for line in open(path):
            if "*" in line:
                return int(line.split(":")[1].strip().lower().replace("mhz", "").replace("*", "").strip())
    except Exception:
        return None
    return None


def main():
    args = sys.argv[1:]
    interval, label = 0.02, "cmd"
    while args and args[0] != "--":
        if args[0] == "--interval":
            interval = float(args[1]); args = args[2:]
        elif args[0] == "--label":
            label = args[1]; args = args[2:]
        else:
            break
    cmd = args[1:] if args and args[0] == "--" else args
    hw = find_hwmon()
    if not hw:
        print(f"[{label}] no hwmon under /sys/class/drm/card*/device: telemetry unavailable on this box")
    card, mon = hw[0] if hw else (None, None)
    files = {}
    if mon:
        for name in ("freq1_input", "power1_average", "power1_input", "temp1_input"):
            p = os.path.join(mon, name)
            if os.path.exists(p):
                files[name] = p
    dpm = os.path.join(card, "pp_dpm_sclk") if card else None
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            s = {"t": time.perf_counter()}
            for k, p in files.items():
                s[k] = read_int(p)
            if dpm:
                s["dpm_sclk_mhz"] = read_dpm(dpm)
            samples.append(s)
            time.sleep(interval)

    idle = {k: read_int(p) for k, p in files.items()}
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.perf_counter()
    rc = subprocess.call(cmd)
    t1 = time.perf_counter()
    stop.set()
    th.join()
    pk = "power1_average" if "power1_average" in files else ("power1_input" if "power1_input" in files else None)
    print(f"[{label}] rc {rc}, {t1 - t0:.2f} s, {len(samples)} samples every {interval * 1e3:.0f} ms from {mon}; idle before: "
          + ", ".join(f"{k} {v}" for k, v in idle.items()))
    if not samples or pk is None:
        return rc
    pw = [s[pk] for s in samples if s.get(pk) is not None]
    thr = (idle.get(pk) or 0) + 0.25 * (max(pw) - (idle.get(pk) or 0)) if pw else 0
    busy = [s for s in samples if (s.get(pk) or 0) >= thr]

    def stat(key, scale, unit):
        v = sorted(s[key] * scale for s in busy if s.get(key) is not None)
        if v:
            print(f"[{label}]   {key:16s} busy samples {len(v):4d}: min {v[0]:8.1f} median {v[len(v) // 2]:8.1f} max {v[-1]:8.1f} {unit}")
    stat("freq1_input", 1e-6, "MHz (hwmon sclk)")
    stat("dpm_sclk_mhz", 1.0, "MHz (pp_dpm_sclk level)")
    stat(pk, 1e-6, "W")
    stat("temp1_input", 1e-3, "C")
    return rc


if __name__ == "__main__":
    sys.exit(main())
