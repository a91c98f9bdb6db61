#!/usr/bin/env python3
"""Shader clock, socket power and junction temperature of GPU 0 sampled while a command runs -- is the bulk of a generation
power-limited (DVFS), which would explain why co-running kernels add up instead of overlapping?
    python tools/clock_watch.py out.csv -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --extra none
Sources: the amdgpu hwmon sysfs files of every card (freq1_input, power1_average|power1_input, temp2_input; the card that worked is kept), else rocm-smi --csv."""
import glob, os, subprocess, sys, time

out, cmd = sys.argv[1], sys.argv[sys.argv.index("--") + 1:]


def hip_pci_bus_id():
    """PCI address of HIP device 0 (the box shows every card of the host in sysfs, the container owns one)"""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
            return buf.value.decode().lower()
    except OSError:
        pass
    return None


def sysfs_sources():
    """one reader per amdgpu hwmon; the card whose PCI address is HIP device 0's when that can be asked, else every card"""
    out = []
    mine = hip_pci_bus_id()
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if mine and os.path.basename(os.path.realpath(os.path.join(hw, "..", ".."))).lower() != mine:
            continue
        f = {k: os.path.join(hw, k) for k in ("freq1_input", "power1_average", "power1_input", "temp2_input", "temp1_input")}
        pw = f["power1_average"] if os.path.exists(f["power1_average"]) else f["power1_input"]
        tp = f["temp2_input"] if os.path.exists(f["temp2_input"]) else f["temp1_input"]
        if os.path.exists(f["freq1_input"]) and os.path.exists(pw):
            def rd(p):
                try:
                    return int(open(p).read())
                except (OSError, ValueError):
                    return -1
            out.append((hw.split("/")[4], lambda f=f, pw=pw, tp=tp, rd=rd: (rd(f["freq1_input"]) / 1e6, rd(pw) / 1e6, rd(tp) / 1e3)))
    return out


def smi_source():
    def rd():
        try:
            o = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp", "--csv"], capture_output=True, text=True, timeout=5).stdout
        except (OSError, subprocess.TimeoutExpired):
            return (-1, -1, -1)
        hdr = row = None
        for l in o.splitlines():
            if l.startswith("device,"):
                hdr = l.split(",")
            elif l.startswith("card"):
                row = l.split(",")
        if not hdr or not row:
            return (-1, -1, -1)
        d = dict(zip(hdr, row))
        g = lambda key: next((v for k, v in d.items() if key in k), "-1")
        return (float(g("sclk clock speed").strip("()Mhz") or -1), float(g("Power") or -1), float(g("junction") or -1))
    return rd


srcs = sysfs_sources()
kind = "sysfs" if srcs else "rocm-smi"
if not srcs:
    srcs = [("card0", smi_source())]
p = subprocess.Popen(cmd)
t0 = time.time()
rows = {name: [] for name, _ in srcs}
while p.poll() is None:
    t = time.time() - t0
    for name, rd in srcs:
        rows[name].append((t,) + tuple(rd()))
    time.sleep(0.02)
best = max(rows, key=lambda n: max((r[1] for r in rows[n]), default=0) - min((r[1] for r in rows[n]), default=0))
with open(out, "w") as f:
    f.write("# source %s, %s (HIP device 0 by PCI address; with several candidates the one whose shader clock moved most)\nt_s,sclk_mhz,power_w,temp_c\n" % (kind, best))
    for r in rows[best]:
        f.write("%.3f,%.0f,%.1f,%.1f\n" % r)
sys.exit(p.returncode)
