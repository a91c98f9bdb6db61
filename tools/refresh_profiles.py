#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of one gpurun_out/<dir> into profiles/ (tracked) and derive profiles/r01_pmc.json.
usage: python tools/refresh_profiles.py gpurun_out/r01d"""
import collections, csv, json, os, shutil, sys
R = sys.argv[1]; P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
d = json.loads(open(R + '/bench_default.json').read().strip().splitlines()[-1])
print("value", d['value'], "ms/step", d['ms_per_step'])
print({k: d['roofline'].get(k) for k in ('achieved', 'frac', 'units_per_launch', 'avg_launch_ms', 'launches', 'traffic', 'traffic_rate')})
print(d['stage_ms_per_generation']); print(d['cpu_baseline'])
shutil.copy(R + '/stats/bench_kernel_stats.csv', P + '/r01_bench_kernel_stats.csv')
shutil.copy(R + '/bench_default.json', P + '/r01_bench_default.json')
shutil.copy(R + '/bench_profiled.json', P + '/r01_bench_under_rocprofv3.json')
def agg(kind, cname):
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(R + '/pmc_%s/kb_counter_collection.csv' % kind)):
        if r['Counter_Name'] == cname:
            by[(r['Kernel_Name'].split('(')[0].replace('void ', ''), int(r.get('Grid_Size', r.get('Grid_Size_X', 0))))].append(float(r['Counter_Value']))
    return {k: (len(v), sum(v) / len(v)) for k, v in by.items()}
f = agg('fetch', 'FETCH_SIZE'); w = agg('write', 'WRITE_SIZE')
lines = ["kernel,grid_size,dispatches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg"]
for k in sorted(set(f) | set(w)):
    lines.append("%s,%d,%d,%.1f,%.1f" % (k[0], k[1], f.get(k, (0, 0))[0], f.get(k, (0, 0))[1], w.get(k, (0, 0))[1]))
open(P + '/r01_pmc_fetch_write_by_kernel.csv', 'w').write("\n".join(lines) + "\n")
kf = max((k for k in f if k[0].startswith('dne::k_fc2<')), key=lambda k: f[k][1])   # the full-width launches (one window: DNE_NSUB=1)
units = 5000   # 2500 pairs = 5000 member-steps per launch
fetch = f[kf][1] * 1024 * 2; write = w[kf][1] * 1024
out = json.load(open(P + '/r01_pmc.json'))
out['k_fc_step'].update({'kernel': kf[0], 'grid_size': kf[1], 'FETCH_SIZE_KB_avg': f[kf][1], 'WRITE_SIZE_KB_avg': w[kf][1],
                         'hbm_bytes_per_launch': fetch + write, 'hbm_bytes_per_unit': (fetch + write) / units})
for name, key in (('k_materialize', 'dne::k_materialize'), ('k_weighted_sum', 'dne::k_weighted_sum')):
    kk = [k for k in f if k[0] == key][0]
    out[name]['FETCH_SIZE_KB'] = f[kk][1]; out[name]['WRITE_SIZE_KB'] = w[kk][1]
json.dump(out, open(P + '/r01_pmc.json', 'w'), indent=1)
print("k_fc hbm bytes per unit", out['k_fc_step']['hbm_bytes_per_unit'])
