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
      dd.get('roofline_ref_pass', {}).get('frac'), dd.get('cpu_baseline', {}).get('value'))
for k, v in dd.get('extra', {}).items():
    print("  extra", k, v.get('value'), (v.get('roofline') or {}).get('frac'), v.get('error'))
for src, dst in (('bench_default.json', 'bench_default.json'), ('bench_driver_cmd.json', 'bench_steps20_warmup5.json'),
                 ('bench_profiled.json', 'bench_under_rocprofv3.json'), ('micro.json', 'micro.json'), ('ga_bench.jsonl', 'ga_bench.jsonl'),
                 ('nses_bench.jsonl', 'nses_bench.jsonl'), ('population_shares.jsonl', 'population_shares.jsonl'),
                 ('len_profile_312.json', 'len_profile_312_pairs.json'), ('len_profile_2500.json', 'len_profile_2500_pairs.json'),
                 ('tail_bench.json', 'tail_bench.json'), ('pmc_mfma/summary.json', 'pmc_mfma.json'),
                 ('ga_large_bench.jsonl', 'ga_large_bench.jsonl'), ('ga_large_kernel_stats.csv', 'ga_large_kernel_stats.csv'),
                 ('six_game_sweep.jsonl', 'six_game_sweep.jsonl'), ('ga_lockstep_profile.json', 'ga_lockstep_profile.json'),
                 ('ga_large_lockstep_profile.json', 'ga_large_lockstep_profile.json'), ('ga_kernel_stats.csv', 'ga_kernel_stats.csv'),
                 ('tail_stats_1.csv', 'tail_stats_1_pair.csv'), ('tail_stats_8.csv', 'tail_stats_8_pairs.csv'), ('tail_stats_24.csv', 'tail_stats_24_pairs.csv'),
                 ('tail_timeline_8.json', 'tail_timeline_8_pairs.json'), ('phase_clock_8.json', 'tail_phase_clock_8_pairs.json'),
                 ('phase_clock_24.json', 'tail_phase_clock_24_pairs.json'), ('ref_bench.json', 'ref_pass.json'),
                 ('ref_bench_one_chunk.json', 'ref_pass_one_chunk.json'), ('ref_pass_one_chunk_kernel_stats.csv', 'ref_pass_one_chunk_kernel_stats.csv'),
                 ('launch_floor.jsonl', 'launch_floor.jsonl'), ('valu_rate.jsonl', 'valu_rate.jsonl'),
                 ('ref_phase_clock.json', 'ref_phase_clock.json'), ('clock_watch_ref.csv', 'clock_watch_ref.csv'),
                 ('clock_watch_bench.csv', 'clock_watch_bench.csv')):
    if os.path.exists(os.path.join(R, src)):
        shutil.copy(os.path.join(R, src), os.path.join(P, '%s_%s' % (PFX, dst)))
ks = glob.glob(R + '/stats/**/*kernel_stats.csv', recursive=True)
if ks:
    shutil.copy(ks[0], P + '/%s_bench_kernel_stats.csv' % PFX)


# the per-regime HBM traffic of the streaming fc kernel (tools/collect_pmc_regimes.sh) -> profiles/<prefix>_pmc.json
import subprocess
if os.path.exists(R + '/%s_pmc.json' % PFX):
    # summarised on the GPU box BEFORE the bench lines were taken (tools/collect_profiles.sh): the very file those lines quote
    shutil.copy(R + '/%s_pmc.json' % PFX, P + '/%s_pmc.json' % PFX)
elif os.path.isdir(R + '/pmc_regimes'):
    subprocess.check_call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'summarize_pmc_regimes.py'), R + '/pmc_regimes', PFX])
