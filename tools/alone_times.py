#!/usr/bin/env python3
"""What each kernel of a full-width lock-step costs ALONE and what it costs in the mix ("written by a tool, not prose").
Inputs: two summaries of tools/trace_summary.py over `rocprofv3 --kernel-trace -- python tools/kbench.py --pairs 2500 --reps 1 --tslimit 8`
  alone: DNE_NSUB=1 -- one window, one stream: every launch has the chip to itself, its duration is the kernel's alone-time for 5000 members
  mix:   the default four windows -- a launch covers a quarter of the members and shares the chip with the other windows' kernels
Output (stdout, JSON): per kernel the alone-time per lock-step, the mix duration per launch and per lock-step (x windows), the stretch, and the
sum of alone-times against the measured lock-step of both runs.
    python tools/alone_times.py gpurun_out/<tag>/es_2500_alone.summary.json gpurun_out/<tag>/es_2500.summary.json > profiles/r06_alone_times.json"""
import json
import sys

alone, mix = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
STEPS = 8
PER_STEP = ("k_fc_ring", "k_conv12", "k_env_render", "k_out", "k_env_logic")


def lockstep_us(d):
    """span of the traced lock-steps / their number (the first k_unit_order and the closing k_compact ride inside the span)"""
    return d["lock_step_span_us"] / STEPS


nwin_mix = len([s for s, v in mix["streams"].items() if v["launches"] > STEPS])
out = {"source": "rocprofv3 --kernel-trace over tools/kbench.py --pairs 2500 --reps 1 --tslimit 8 (nobody dies: every lock-step is 5000 members wide); "
                 "alone = DNE_NSUB=1 (one window), mix = the default schedule (%d windows)" % nwin_mix,
       "lock_step_us": {"alone_schedule": round(lockstep_us(alone), 1), "mix": round(lockstep_us(mix), 1)},
       "concurrency_mix": mix["sum_of_durations_over_span"], "kernels": {}}
tot = 0.0
for k, v in alone["kernels"].items():
    if not k.startswith(PER_STEP):
        continue
    m = mix["kernels"].get(k)
    row = {"alone_us_per_lock_step": v["dur_us_mean"], "gap_before_us_alone": v["gap_before_us_mean"]}
    tot += v["dur_us_mean"]
    if m:
        row.update({"mix_us_per_launch": m["dur_us_mean"], "mix_us_per_lock_step_all_windows": round(m["dur_us_mean"] * nwin_mix, 1),
                    "stretch_in_mix": round(m["dur_us_mean"] * nwin_mix / v["dur_us_mean"], 2)})
    out["kernels"][k] = row
out["sum_of_alone_us"] = round(tot, 1)
out["mix_lock_step_over_sum_of_alone"] = round(lockstep_us(mix) / tot, 3)
print(json.dumps(out, indent=1))
