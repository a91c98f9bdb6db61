#!/usr/bin/env python3
"""Copy the summaries of one gpurun_out/<dir> (tools/collect_profiles.sh) into profiles/ (tracked) under a round prefix and
derive profiles/<prefix>_pmc.json.   usage: python tools/refresh_profiles.py gpurun_out/r02p r02"""
import collections, csv, glob, json, os, shutil, sys
R, PFX = sys.argv[1], sys.argv[2]
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


d = last_json(R + '/bench_default.json')
print("default: value", d['value'], "ms/step", d['ms_per_step'])
dd = last_json(R + '/bench_driver_cmd.json')
print("driver cmd: value", dd['value'], "ms/step", dd['ms_per_step'], {k: dd['roofline'].get(k) for k in ('frac', 'frac_counter', 'avg_launch_ms', 'launches')},
      dd.get('roofline_ref_pass', {}).get('frac'), dd['cpu_baseline']['value'])
for src, dst in (('bench_default.json', 'bench_default.json'), ('bench_driver_cmd.json', 'bench_steps20_warmup5.json'),
                 ('bench_profiled.json', 'bench_under_rocprofv3.json'), ('micro.json', 'micro.json'), ('ga_bench.jsonl', 'ga_bench.jsonl'),
                 ('nses_bench.jsonl', 'nses_bench.jsonl'), ('population_shares.jsonl', 'population_shares.jsonl'),
                 ('len_profile_312.json', 'len_profile_312_pairs.json'), ('len_profile_2500.json', 'len_profile_2500_pairs.json'),
                 ('tail_bench.json', 'tail_bench.json'), ('pmc_mfma/summary.json', 'pmc_mfma.json'),
                 ('ga_large_bench.jsonl', 'ga_large_bench.jsonl'), ('ga_large_kernel_stats.csv', 'ga_large_kernel_stats.csv'),
                 ('six_game_sweep.jsonl', 'six_game_sweep.jsonl')):
    if os.path.exists(os.path.join(R, src)):
        shutil.copy(os.path.join(R, src), os.path.join(P, '%s_%s' % (PFX, dst)))
ks = glob.glob(R + '/stats/**/*kernel_stats.csv', recursive=True)
if ks:
    shutil.copy(ks[0], P + '/%s_bench_kernel_stats.csv' % PFX)


def agg(kind, cname):
    by = collections.defaultdict(lambda: collections.defaultdict(float))
    f = glob.glob(R + '/pmc_%s/**/*counter_collection.csv' % kind, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == cname:
            k = (r['Kernel_Name'].split('(')[0].replace('void ', ''), int(r.get('Grid_Size', r.get('Grid_Size_X', 0))))
            by[k][r['Dispatch_Id']] += float(r['Counter_Value'])
    return {k: (len(v), sum(v.values()) / len(v)) for k, v in by.items()}


f = agg('fetch', 'FETCH_SIZE'); w = agg('write', 'WRITE_SIZE')
lines = ["kernel,grid_size,dispatches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg"]
for k in sorted(set(f) | set(w)):
    lines.append("%s,%d,%d,%.1f,%.1f" % (k[0], k[1], f.get(k, (0, 0))[0], f.get(k, (0, 0))[1], w.get(k, (0, 0))[1]))
open(P + '/%s_pmc_fetch_write_by_kernel.csv' % PFX, 'w').write("\n".join(lines) + "\n")
kf = max((k for k in f if k[0].startswith('dne::k_fc_duo<') or k[0].startswith('dne::k_fc2<')), key=lambda k: f[k][1])   # the full-width launches (one window: DNE_NSUB=1)
units = 5000   # 2500 pairs = 5000 member-steps per launch
fetch = f[kf][1] * 1024 * 2; write = w[kf][1] * 1024
out = {"k_fc_step": {"kernel": kf[0], "grid_size": kf[1], "units_per_launch": units, "FETCH_SIZE_KB_avg": f[kf][1], "WRITE_SIZE_KB_avg": w[kf][1],
                     "fetch_correction": "x2 (16 B/lane streaming loads, per MI355X_MICROARCH.md; the uncorrected value would be below the noise bytes that must be read)",
                     "hbm_bytes_per_launch": fetch + write, "hbm_bytes_per_unit": (fetch + write) / units,
                     "algorithmic_bytes_per_unit": 4064456, "noise_bytes_per_pair": 3964928},
       "command": "DNE_NSUB=1 rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python tools/kbench.py --reps 1 --tslimit 6  "
                  "(separate passes, 2500 pairs in one window: a full-width fc launch = 5000 member-steps)"}
for name, key, note in (('k_materialize', 'dne::k_materialize', "4 B/lane loads: FETCH_SIZE matches the known byte count without correction (calibration point)"),
                        ('k_weighted_sum', 'dne::k_weighted_sum', "4 B/lane loads; the 1 GB table is re-read ~10x so part of the traffic is served by L2 / Infinity Cache")):
    kk = [k for k in f if k[0] == key]
    if kk:
        out[name] = {"FETCH_SIZE_KB": f[kk[0]][1], "WRITE_SIZE_KB": w.get(kk[0], (0, 0))[1], "note": note}
json.dump(out, open(P + '/%s_pmc.json' % PFX, 'w'), indent=1)
print("k_fc hbm bytes per unit", out['k_fc_step']['hbm_bytes_per_unit'])
