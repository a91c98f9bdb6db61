#!/usr/bin/env python3
"""Shader clock, socket power and junction temperature of GPU 0 sampled while a command runs -- is the bulk of a generation
power-limited (DVFS), which would explain why co-running kernels add up instead of overlapping?
    python tools/clock_watch.py out.csv -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --extra none
Sources, first that works: the amdgpu hwmon sysfs files (freq1_input, power1_average|power1_input, temp2_input), else rocm-smi --csv."""
import glob, os, subprocess, sys, time

out, cmd = sys.argv[1], sys.argv[sys.argv.index("--") + 1:]


def sysfs_source():
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        f = {k: os.path.join(hw, k) for k in ("freq1_input", "power1_average", "power1_input", "temp2_input", "temp1_input")}
        pw = f["power1_average"] if os.path.exists(f["power1_average"]) else f["power1_input"]
        tp = f["temp2_input"] if os.path.exists(f["temp2_input"]) else f["temp1_input"]
        if os.path.exists(f["freq1_input"]) and os.path.exists(pw):
            def rd(p):
                try:
                    return int(open(p).read())
                except (OSError, ValueError):
                    return -1
            return lambda: (rd(f["freq1_input"]) / 1e6, rd(pw) / 1e6, rd(tp) / 1e3)
    return None


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


src = sysfs_source()
kind = "sysfs" if src else "rocm-smi"
src = src or smi_source()
p = subprocess.Popen(cmd)
t0 = time.time()
with open(out, "w") as f:
    f.write("# source %s\nt_s,sclk_mhz,power_w,temp_c\n" % kind)
    while p.poll() is None:
        s = src()
        f.write("%.3f,%.0f,%.1f,%.1f\n" % (time.time() - t0, *s))
        f.flush()
        time.sleep(0.02)
sys.exit(p.returncode)
