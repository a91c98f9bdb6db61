#!/usr/bin/env python3
"""What the host under this process can actually give a CPU baseline: os.cpu_count() counts the machine's logical CPUs, not the
ones this process may run on (affinity mask) nor the share a container's cgroup grants (cpu.max / cfs quota).  bench.py's
`cpu_baseline` and the full-generation parity tests size their worker pools from usable_cpus() and report host_facts()."""
import math
import os
import re


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def cgroup_cpu_limit():
    """CPUs' worth of time the cgroup grants per period (float), or None when unlimited / unknown.  cgroup v2: cpu.max =
    '<quota> <period>' or 'max <period>'; v1: cpu.cfs_quota_us / cpu.cfs_period_us (-1 = unlimited)."""
    v2 = _read("/sys/fs/cgroup/cpu.max")
    if v2:
        parts = v2.split()
        if len(parts) == 2 and parts[0] != "max":
            try:
                return float(parts[0]) / float(parts[1])
            except ValueError:
                return None
        return None
    q, p = _read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), _read("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
    try:
        if q is not None and p is not None and float(q) > 0:
            return float(q) / float(p)
    except ValueError:
        pass
    return None


def affinity_cpus():
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def usable_cpus():
    """worker processes that can run at the same time: the affinity mask, cut to the cgroup's CPU quota"""
    n = affinity_cpus()
    lim = cgroup_cpu_limit()
    if lim is not None:
        n = min(n, max(1, int(math.ceil(lim))))
    return max(1, n)


def cpu_model():
    info = _read("/proc/cpuinfo") or ""
    m = re.search(r"model name\s*:\s*(.+)", info)
    sockets = len(set(re.findall(r"physical id\s*:\s*(\d+)", info))) or None
    cores = set(re.findall(r"physical id\s*:\s*(\d+)[\s\S]*?core id\s*:\s*(\d+)", info))
    return {"model": m.group(1).strip() if m else None, "sockets": sockets, "physical_cores": len(cores) or None}


def host_facts():
    lim = cgroup_cpu_limit()
    d = {"os_cpu_count": os.cpu_count(), "sched_getaffinity": affinity_cpus(), "cgroup_cpu_max": _read("/sys/fs/cgroup/cpu.max"),
         "cgroup_cpu_limit": lim, "usable_cpus": usable_cpus(), "loadavg": _read("/proc/loadavg")}
    d.update(cpu_model())
    mem = _read("/proc/meminfo") or ""
    m = re.search(r"MemTotal:\s*(\d+) kB", mem)
    d["mem_total_gb"] = round(int(m.group(1)) / 2 ** 20, 1) if m else None
    return d


if __name__ == "__main__":
    import json
    print(json.dumps(host_facts(), indent=1))
