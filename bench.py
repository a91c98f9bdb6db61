#!/usr/bin/env python3
"""bench.py -- env-steps/sec/generation of the ES hot path (BASELINE.json metric) on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one ES generation of the workload named in config.workload: sample the noise indices, evaluate
every antithetic pair of the population (reset, virtual-batch-norm reference pass, lock-step act -> env.step
until each episode ends), exchange (noise_idx, return) records between GPUs, centered-rank -> weighted noise
sum -> Adam on every rank.  The population (pop 5000 = 2500 pairs) is sharded round-robin across the N GPUs
("strong" scaling: total work is fixed, as the metric "Frostbite ES pop 5000 at 1/2/4/8 GPUs" states).
Inputs (noise table, theta, reference batch) are resident in HBM before the timed region.
value = sum of episode lengths over all ranks and timed generations / max-over-ranks wall time (es.py:332,341).

Environment: ALE and ROMs do not exist in this image, so the emulator under wrap_deepmind is the
Frostbite-shaped SynthAtari fixture -- stated in "data".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))

ALG_BYTES_PER_ENV_STEP = 4 * 1009058 + 28224   # all member weights once + the u8 observation stack
HBM_PEAK = 8.0e12                              # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FC_KERNELS = {   # the kernel behind the profiled ("full") fc launches of one evaluation (dne_profile.fc_full_kind)
    2: "dne::k_fc2<true, 4> (streaming fc + bn + out + argmax, two antithetic pairs per work item: every lock-step with "
       ">= 800 active pairs on the rank)",
    1: "dne::k_fc<2, false, true, 4> (streaming fc + bn + out + argmax, one pair per work item: every window with > 96 active "
       "pairs; the rank's share is too small for k_fc2)",
}

EXP = {
    "config": {"calc_obstat_prob": 0.0, "episodes_per_batch": 5000, "eval_prob": 0.0, "l2coeff": 0.005,
               "noise_stdev": 0.02, "snapshot_freq": 0, "timesteps_per_batch": 10000,
               "return_proc_mode": "centered_rank", "episode_cutoff_mode": 5000},
    "env_id": "FrostbiteNoFrameskip-v4",
    "optimizer": {"args": {"stepsize": 0.01}, "type": "adam"},
    "policy": {"args": {}, "type": "ESAtariPolicy"},
}


def cpu_baseline(noise, theta, ref, sigma, tslimit, n_actions):
    """The CPU oracle, structured like the reference workers (one single-threaded process per core, one
    antithetic pair at a time, batch-1 forwards, a reference pass per episode: es.py:366-439, launch.py:117),
    timed on a bounded sample of generation 0 of the same workload.  Checker only -- never the product path.
    The reference's workers never idle (they loop over tasks), so the rate is steps per busy core-second
    times the core count: sum(steps) / (sum(worker busy seconds) / cores)."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.lib()
    cores = os.cpu_count() or 1
    from dne_hip import es
    _, idx, seeds = es.generation_inputs(noise.size, theta.size, 2500, 0, 0, 1)
    n = min(2 * cores, 2500)
    global _BASE
    _BASE = (noise, theta, ref, sigma, tslimit, n_actions, idx, seeds)
    ctx = mp.get_context("fork")
    t0 = time.time()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_pair, range(n), chunksize=1)
    wall = time.time() - t0
    steps = int(sum(r[0] for r in res))
    busy = float(sum(r[1] for r in res))
    return {"value": steps / (busy / cores), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "generation 0, first %d antithetic pairs (%d full episodes, %d env-steps) over %d single-threaded "
                      "worker processes; %.1f busy core-seconds, %.1f s wall (wall-clock rate incl. stragglers and "
                      "process start-up: %.0f steps/s)" % (n, 2 * n, steps, cores, busy, wall, steps / wall)}


def _cpu_pair(i):
    os.environ["OMP_NUM_THREADS"] = "1"
    import oracle as O
    noise, theta, ref, sigma, tslimit, n_actions, idx, seeds = _BASE
    L = O.layout(O.KIND_ES, n_actions)
    t0 = time.time()
    _, _, ln = O.es_eval(L, theta, noise, idx[i:i + 1], sigma, tslimit, ref, seeds[2 * i:2 * i + 2])
    return int(ln.sum()), time.time() - t0


def _pmc_traffic(units_per_launch):
    """HBM bytes per launch of the fc kernel from the committed rocprofv3 PMC passes (profiles/r01_pmc.json:
    FETCH_SIZE doubled for the 16-byte streaming loads as MI355X_MICROARCH.md prescribes, plus WRITE_SIZE),
    measured per unit at full width and scaled to this run's units per launch.  None if no profile is committed."""
    p = os.path.join(ROOT, "profiles", "r01_pmc.json")
    if not os.path.exists(p):
        return None
    try:
        per_unit = json.load(open(p))["k_fc_step"]["hbm_bytes_per_unit"]
        return per_unit * units_per_launch
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pop", type=int, default=5000)
    ap.add_argument("--tslimit", type=int, default=5000)
    ap.add_argument("--noise-count", type=int, default=250_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-events", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "exercise the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--single-device", action="store_true", help="testing aid: every rank uses GPU 0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)" % (args.gpus, world))

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    gather_device = device if args.backend == "nccl" else None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from dne_hip import _lib, es, policies
    n_pairs = args.pop // 2
    config = es.Config(**EXP["config"])
    my_pairs = len(es.shard_pairs(n_pairs, rank, world))
    engine = _lib.Engine(_lib.KIND_ES, 18, max_members=2 * my_pairs, ref_count=128, device_id=local_rank,
                         profile_events=not args.no_profile_events)
    t0 = time.time()
    noise = es.SharedNoiseTable(count=args.noise_count)
    noise.attach(engine)
    t_noise = time.time() - t0
    theta0 = policies.xavier_flat(18, seed=0)
    engine.set_theta(theta0)
    env = policies.HipAtariEnv(engine, seed=0)
    ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
    engine.set_ref_batch(ref)
    engine.optimizer_reset()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    gen = 0
    for _ in range(args.warmup):
        es.es_generation(engine, noise.noise.size, config, n_pairs, gen, args.tslimit, EXP["optimizer"], rank, world, gather_device)
        gen += 1
    barrier()
    t0 = time.time()
    steps_local = 0
    fc_ms = fc_launches = fc_units = 0
    fc_all_ms = 0.0
    fc_kind = 2
    fc_union_ms = 0.0
    stage = {"conv_ms": 0.0, "env_ms": 0.0, "ref_ms": 0.0, "reduce_ms": 0.0, "eval_ms": 0.0}
    for _ in range(args.steps):
        rec, ratio = es.es_generation(engine, noise.noise.size, config, n_pairs, gen, args.tslimit, EXP["optimizer"], rank, world, gather_device)
        gen += 1
        p = engine.profile()
        steps_local += p["env_steps"]
        # roofline kernel = the k_fc2 launches only (mid-range and tail lock-steps run other fc kernels)
        fc_ms += p["fc_full_ms"]; fc_launches += p["fc_full_launches"]; fc_units += p["fc_full_units"]
        fc_all_ms += p["fc_ms"]; fc_kind = int(p["fc_full_kind"]); fc_union_ms += p["fc_full_union_ms"]
        for k in stage:
            stage[k] += p[k]
    barrier()
    wall = time.time() - t0
    tot = torch.tensor([float(steps_local), wall], dtype=torch.float64, device=gather_device if gather_device is not None else "cpu")
    if world > 1:
        steps_t = tot[0:1].clone(); wall_t = tot[1:2].clone()
        dist.all_reduce(steps_t, op=dist.ReduceOp.SUM)
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
        total_steps, wall = float(steps_t.item()), float(wall_t.item())
    else:
        total_steps = float(steps_local)
    theta_sum = float(np.abs(engine.get_theta()).sum())

    if rank == 0:
        value = total_steps / wall
        out = {
            "metric": "env-steps/sec/generation (Frostbite ES pop 5000)", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * wall / max(args.steps, 1),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic: Frostbite-shaped SynthAtari fixture under wrap_deepmind (ALE/ROMs unavailable); "
                    "reference noise table seed 123 x %d; Xavier theta RandomState(0); 128-frame reference batch" % args.noise_count,
            "config": {"workload": "FrostbiteNoFrameskip-v4 ES pop=%d (N=%d antithetic pairs), Nature-CNN ESAtariPolicy "
                                   "(P=1009058, virtual batch norm over 128 reference frames), 84x84x4 u8, sigma=0.02, "
                                   "tslimit=%d, centered_rank + Adam(0.01) + l2 0.005" % (args.pop, n_pairs, args.tslimit),
                       "pairs_per_gpu": my_pairs, "parallelism": "population sharded round-robin over %d GPU(s), "
                                                                 "all-gather of 32-byte records, redundant update" % world},
        }
        if fc_ms > 0:
            avg_ms = fc_ms / fc_launches
            units_per_launch = fc_units / fc_launches
            achieved = units_per_launch * ALG_BYTES_PER_ENV_STEP / (avg_ms * 1e-3)
            out["roofline"] = {
                "bound": "hbm", "kernel": FC_KERNELS[fc_kind],
                "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
                "traffic": _pmc_traffic(units_per_launch),
                "algorithmic_bytes_per_unit": ALG_BYTES_PER_ENV_STEP, "unit_def": "one env-step of one member",
                "units_per_launch": units_per_launch, "avg_launch_ms": avg_ms, "launches": int(fc_launches),
                "note": "antithetic pairs share one read of their noise slice, so the HBM traffic per unit (traffic, from "
                        "the rocprofv3 FETCH_SIZE/WRITE_SIZE passes in profiles/) is about half the algorithmic figure and "
                        "frac may exceed 1; traffic_rate is what the memory system actually delivers to this kernel",
            }
            # SURVEY 8d also asks for the whole-job figure: every env-step of the generation (reference pass, tail and
            # update included in the time) priced at the same algorithmic bytes
            out["roofline"]["whole_job"] = {"achieved": value * ALG_BYTES_PER_ENV_STEP / 1e9, "unit": "GB/s",
                                            "frac": value * ALG_BYTES_PER_ENV_STEP / (HBM_PEAK * world)}
            # the windows launch this kernel concurrently from several streams; a launch that shares the chip with its siblings is
            # stretched, so also: all profiled bytes / the time during which at least one such launch was running
            if fc_union_ms > 0:
                uni = fc_units * ALG_BYTES_PER_ENV_STEP / (fc_union_ms * 1e-3)
                out["roofline"]["concurrent_launches"] = {"achieved": uni / 1e9, "unit": "GB/s", "frac": uni / HBM_PEAK,
                                                          "busy_ms_per_generation": fc_union_ms / args.steps}
            tr = out["roofline"]["traffic"]
            if tr:
                out["roofline"]["traffic_rate"] = {"value": tr / (avg_ms * 1e-3) / 1e9, "unit": "GB/s",
                                                   "frac_of_peak": tr / (avg_ms * 1e-3) / HBM_PEAK}
            out["stage_ms_per_generation"] = {k: v / args.steps for k, v in stage.items()}
            out["stage_ms_per_generation"]["fc_ms"] = fc_all_ms / args.steps
            out["stage_ms_per_generation"]["fc_streaming_kernel_ms"] = fc_ms / args.steps
        else:
            # this rank's share never reaches the streaming kernels' range (e.g. 312 pairs at N = 8 run in windows of <= 96 pairs
            # on the column-split kernels, which are not bracketed by events): only the whole-job figure is available
            per_gpu = value * ALG_BYTES_PER_ENV_STEP / world
            out["roofline"] = {"bound": "hbm", "kernel": "none profiled (all lock-steps in the column-split / quad fc range)",
                               "achieved": per_gpu / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": per_gpu / HBM_PEAK,
                               "traffic": None, "algorithmic_bytes_per_unit": ALG_BYTES_PER_ENV_STEP,
                               "unit_def": "one env-step of one member", "note": "whole-job algorithmic bytes per GPU / wall time",
                               "whole_job": {"achieved": value * ALG_BYTES_PER_ENV_STEP / 1e9, "unit": "GB/s",
                                             "frac": value * ALG_BYTES_PER_ENV_STEP / (HBM_PEAK * world)}}
            out["stage_ms_per_generation"] = {k: v / args.steps for k, v in stage.items()}
        out["setup_s"] = {"noise_table": t_noise}
        out["theta_abs_sum_after"] = theta_sum
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(noise.noise, theta0, ref, config.noise_stdev, args.tslimit, 18)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    engine.close()


if __name__ == "__main__":
    main()
