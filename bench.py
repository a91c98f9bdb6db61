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

This process never imports torch: the engine is reached through ctypes, device synchronisation is
hipDeviceSynchronize inside the C ABI, and for N > 1 the exchange, the barrier and the max/sum over ranks are
RCCL calls behind dne_comm_* (the ranks find each other through RANK / WORLD_SIZE / MASTER_PORT of the launcher;
the 128-byte RCCL id travels through a file in /tmp, all ranks being on one node).  So the HIP runtime in this
process is the one libdne_hip.so was built against (/opt/rocm), the same one the GPU tests run on.

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
MFMA_F32_PEAK = 157.3e12                       # MI355X_MICROARCH.md: dense fp32 MFMA
# reference pass (virtual batch norm): flops of one reference frame through one member's network (2 * MACs)
REF_FLOP_PER_FRAME = 2 * (441 * 256 * 16 + 121 * 256 * 32 + 3872 * 256)
FC_KERNELS = {   # the kernel behind the profiled ("full") fc launches of one evaluation (dne_profile.fc_full_kind)
    3: "dne::k_fc_duo<2, true> (table-ordered streaming fc: a wave takes two (pair, k-slice) units that are neighbours in the noise "
       "table and walks them in table lock-step, 8 rows per stream in flight, so rows both units need reach HBM once; every window "
       "of a lock-step with >= 800 active pairs on the rank; bn3 + output layer + argmax follow in k_out, outside the timed bracket)",
    2: "dne::k_fc2<true, 4> (streaming fc + bn + out + argmax, two antithetic pairs per work item: every lock-step with "
       ">= 800 active pairs on the rank)",
    1: "dne::k_fc<2, false, true, 4> (streaming fc + bn + out + argmax, one pair per work item: every window with > 96 active "
       "pairs; the rank's share is too small for k_fc2)",
}
PMC_PROFILE = os.path.join("profiles", "r02_pmc.json")

EXP = {
    "config": {"calc_obstat_prob": 0.0, "episodes_per_batch": 5000, "eval_prob": 0.0, "l2coeff": 0.005,
               "noise_stdev": 0.02, "snapshot_freq": 0, "timesteps_per_batch": 10000,
               "return_proc_mode": "centered_rank", "episode_cutoff_mode": 5000},
    "env_id": "FrostbiteNoFrameskip-v4",
    "optimizer": {"args": {"stepsize": 0.01}, "type": "adam"},
    "policy": {"args": {}, "type": "ESAtariPolicy"},
}

_T0 = time.time()
_RANK = int(os.environ.get("RANK", "0"))


def crumb(msg):
    """stage breadcrumb on stderr: a device fault kills the process without a Python traceback, the last line says where"""
    sys.stderr.write("[bench r%d %7.2fs] %s\n" % (_RANK, time.time() - _T0, msg))
    sys.stderr.flush()


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


FC_KERNEL_TAG = {3: "k_fc_duo", 2: "k_fc2", 1: "k_fc<"}


def _pmc_traffic(kind):
    """HBM bytes per env-step of the streaming fc kernel from the committed rocprofv3 PMC passes (FETCH_SIZE doubled for the
    16-byte streaming loads as MI355X_MICROARCH.md prescribes, plus WRITE_SIZE; collected by tools/collect_profiles.sh in
    separate --pmc runs of a full-width workload).  It is a property of the kernel measured under the profiler, NOT a
    measurement of this run: bench.py cannot read hardware counters in-process.  (None, None) if no profile is committed."""
    for rel in (PMC_PROFILE, os.path.join("profiles", "r01_pmc.json")):
        p = os.path.join(ROOT, rel)
        if os.path.exists(p):
            try:
                k = json.load(open(p))["k_fc_step"]
                if FC_KERNEL_TAG.get(kind, "?") in k["kernel"]:   # counters of another kernel say nothing about this one
                    return float(k["hbm_bytes_per_unit"]), rel
            except Exception:
                pass
    return None, None


def rccl_rendezvous(_lib, engine, rank, world):
    """All ranks are children of one launcher on one node: rank 0 draws the RCCL id and publishes it through a file
    keyed by the launcher's pid and MASTER_PORT; the others wait for it.  Then every rank joins the communicator."""
    path = os.environ.get("DNE_RCCL_ID_FILE") or "/tmp/dne_rccl_id.%s.%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
    if rank == 0:
        try:
            uid = _lib.comm_unique_id()
        except _lib.DneError:
            uid = None
        with open(path + ".tmp", "wb") as f:
            f.write(uid if uid is not None else b"!")   # a one-byte file tells the waiting ranks that there will be no id
        os.replace(path + ".tmp", path)
        if uid is None:
            raise _lib.DneError("rank 0 could not open RCCL")
    else:
        deadline = time.time() + 600
        while True:
            try:
                uid = open(path, "rb").read()
                if len(uid) == 128:
                    break
                if len(uid) == 1:
                    raise _lib.DneError("rank 0 could not open RCCL")
            except OSError:
                pass
            if time.time() > deadline:
                raise SystemExit("rank %d: no RCCL id at %s after 600 s" % (rank, path))
            time.sleep(0.01)
    engine.comm_init(rank, world, uid)
    engine.barrier()
    if rank == 0:
        try:
            os.unlink(path)
        except OSError:
            pass


def supervise():
    """N = 1 only: run the benchmark proper in a child process.  A GPU memory-access fault is raised by the HSA runtime as an
    abort of the whole process -- no exception, no JSON (the fate of round 1's driver run).  The child's breadcrumbs name the
    stage it died in; one retry with the engine's own trace on (DNE_TRACE=1) then either yields a complete, separately timed
    run -- reported with "attempts": 2 and the first attempt's last stage -- or a second, more detailed failure."""
    import subprocess
    import threading
    first = None
    for attempt in (1, 2):
        env = dict(os.environ, DNE_BENCH_CHILD="1")
        if attempt == 2:
            env["DNE_TRACE"] = "1"
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True)
        tail = []

        def pump():
            for line in p.stderr:
                sys.stderr.write(line); sys.stderr.flush()
                tail.append(line.rstrip())
                del tail[:-6]
        t = threading.Thread(target=pump, daemon=True)
        t.start()
        out = p.stdout.read()
        rc = p.wait()
        t.join(5)
        if rc == 0:
            if first is not None:
                lines = out.strip().splitlines()
                try:
                    d = json.loads(lines[-1])
                    d["attempts"], d["first_attempt"] = 2, first
                    lines[-1] = json.dumps(d)
                    out = "\n".join(lines) + "\n"
                except Exception:
                    pass
            sys.stdout.write(out); sys.stdout.flush()
            return 0
        if rc > 0:      # an ordinary Python failure (bad arguments, no GPU): nothing a retry would change
            sys.stdout.write(out)
            return rc
        first = {"rc": rc, "stderr_tail": tail[-6:]}
        sys.stderr.write("[bench supervisor] attempt %d ended with rc %d; last lines: %s\n" % (attempt, rc, " | ".join(tail[-3:])))
        sys.stderr.flush()
    return rc if rc > 0 else 128 - rc


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
    ap.add_argument("--transport", default="rccl", choices=["rccl", "gloo"],
                    help="exchange for N > 1: rccl = dne_comm_* (RCCL over xGMI behind the C ABI); gloo = torch.distributed on the "
                         "host, only to exercise the multi-rank path on a box with fewer GPUs than ranks")
    ap.add_argument("--single-device", action="store_true", help="testing aid: every rank uses GPU 0 (needs --transport gloo)")
    ap.add_argument("--no-supervisor", action="store_true", help="N = 1: run in this process (no child, no retry)")
    args = ap.parse_args()
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_supervisor and not os.environ.get("DNE_BENCH_CHILD"):
        sys.exit(supervise())

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)" % (args.gpus, world))
    if args.single_device:
        local_rank = 0

    crumb("start: world %d, steps %d, warmup %d" % (world, args.steps, args.warmup))
    from dne_hip import _lib, es, policies   # loads libdne_hip.so (and with it /opt/rocm's HIP runtime) first
    n_pairs = args.pop // 2
    config = es.Config(**EXP["config"])
    my_pairs = len(es.shard_pairs(n_pairs, rank, world))
    try:
        engine = _lib.Engine(_lib.KIND_ES, 18, max_members=2 * my_pairs, ref_count=128, device_id=local_rank,
                             profile_events=not args.no_profile_events)
    except _lib.DneError as e:
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU fallback (%s)" % e)
    crumb("engine created on device %d (%d pairs)" % (local_rank, my_pairs))
    transport = None
    use_gloo = world > 1 and args.transport == "gloo"
    if world > 1 and args.transport == "rccl":
        try:
            rccl_rendezvous(_lib, engine, rank, world)
            crumb("RCCL communicator ready")
        except _lib.DneError as e:
            # ncclCommInitRank fails on every rank or on none (same library, same topology): all ranks take the host path
            # together.  The records are 32 bytes per pair, so the carrier decides nothing about the result, only ~1 ms of latency.
            crumb("RCCL communicator could not be created (%s): falling back to the gloo carrier for the 32-byte records" % e)
            use_gloo = True
    if use_gloo:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        transport = es.allgather_records
    t0 = time.time()
    noise = es.SharedNoiseTable(count=args.noise_count)
    crumb("noise table sampled (%d floats)" % noise.noise.size)
    noise.attach(engine)
    t_noise = time.time() - t0
    crumb("noise table uploaded")
    theta0 = policies.xavier_flat(18, seed=0)
    engine.set_theta(theta0)
    env = policies.HipAtariEnv(engine, seed=0)
    ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
    engine.set_ref_batch(ref)
    engine.optimizer_reset()
    crumb("theta + reference batch set")

    def barrier():
        if transport is not None:
            dist.barrier()
        engine.barrier()    # RCCL all-reduce when a communicator exists, then hipDeviceSynchronize

    gen = 0
    for _ in range(args.warmup):
        crumb("generation %d (warmup) eval" % gen)
        es.es_generation(engine, noise.noise.size, config, n_pairs, gen, args.tslimit, EXP["optimizer"], rank, world, transport)
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
        crumb("generation %d eval" % gen)
        rec, ratio = es.es_generation(engine, noise.noise.size, config, n_pairs, gen, args.tslimit, EXP["optimizer"], rank, world, transport)
        gen += 1
        p = engine.profile()
        steps_local += p["env_steps"]
        # roofline kernel = the full-width streaming launches only (mid-range and tail lock-steps run other fc kernels)
        fc_ms += p["fc_full_ms"]; fc_launches += p["fc_full_launches"]; fc_units += p["fc_full_units"]
        fc_all_ms += p["fc_ms"]; fc_kind = int(p["fc_full_kind"]); fc_union_ms += p["fc_full_union_ms"]
        for k in stage:
            stage[k] += p[k]
    barrier()
    wall = time.time() - t0
    crumb("timed region done: %.3f s" % wall)
    if world > 1 and transport is None:
        total_steps = float(engine.comm_allreduce([float(steps_local)], "sum")[0])
        wall = float(engine.comm_allreduce([wall], "max")[0])
    elif world > 1:
        import torch
        st = torch.tensor([float(steps_local)], dtype=torch.float64); wt = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(st, op=dist.ReduceOp.SUM); dist.all_reduce(wt, op=dist.ReduceOp.MAX)
        total_steps, wall = float(st.item()), float(wt.item())
    else:
        total_steps = float(steps_local)
    theta_final = engine.get_theta()
    theta_sum = float(np.abs(theta_final).sum())
    import hashlib
    crumb("theta sha256 %s" % hashlib.sha256(theta_final.tobytes()).hexdigest())   # identical on every rank (redundant update)
    engine.check_redzones()   # raises if any kernel of the run wrote outside its device buffer
    crumb("red zones intact")

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
                                                                 "RCCL all-gather of 32-byte records, redundant update" % world},
        }
        per_unit, src = _pmc_traffic(fc_kind)
        if fc_ms > 0:
            avg_ms = fc_ms / fc_launches
            units_per_launch = fc_units / fc_launches
            achieved = units_per_launch * ALG_BYTES_PER_ENV_STEP / (avg_ms * 1e-3)
            traffic = per_unit * units_per_launch if per_unit else None
            out["roofline"] = {
                "bound": "hbm", "kernel": FC_KERNELS[fc_kind],
                "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
                "frac_algorithmic": achieved / HBM_PEAK,
                "frac_counter": (traffic / (avg_ms * 1e-3) / HBM_PEAK) if traffic else None,
                "traffic": traffic,
                "traffic_source": ("%s: PMC bytes per env-step of this kernel from separate rocprofv3 --pmc passes (FETCH_SIZE x2 + "
                                   "WRITE_SIZE), scaled to this run's units per launch -- not measured in this run" % src) if src else None,
                "algorithmic_bytes_per_unit": ALG_BYTES_PER_ENV_STEP, "unit_def": "one env-step of one member",
                "units_per_launch": units_per_launch, "avg_launch_ms": avg_ms, "launches": int(fc_launches),
                "note": "frac = frac_algorithmic = SURVEY 8d bytes (every member's weights once per env-step) / launch time / 8 TB/s; "
                        "an antithetic pair shares one read of its noise slice and (k_fc_duo) neighbouring units share table rows, so "
                        "the bytes the memory system actually moves (frac_counter) are well below that -- frac_counter is the honest "
                        "distance to the HBM roofline, and it falls when the kernel avoids traffic",
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
                                                          "frac_counter": (uni / ALG_BYTES_PER_ENV_STEP * per_unit / HBM_PEAK) if per_unit else None,
                                                          "busy_ms_per_generation": fc_union_ms / args.steps}
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
        if fc_ms > 0:
            out["stage_ms_per_generation"]["fc_ms"] = fc_all_ms / args.steps
            out["stage_ms_per_generation"]["fc_streaming_kernel_ms"] = fc_ms / args.steps
        if stage["ref_ms"] > 0:   # the matrix-core phase of a generation, against the dense fp32 MFMA peak (rank 0's share)
            flop = 2 * my_pairs * 128 * REF_FLOP_PER_FRAME
            rate = flop / (stage["ref_ms"] / args.steps * 1e-3)
            out["roofline_ref_pass"] = {"bound": "mfma", "kernels": "k_conv1_ref + k_conv2 + k_fc_ref (+ batch statistics)",
                                        "achieved": rate / 1e12, "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
                                        "frac": rate / MFMA_F32_PEAK, "flop_per_generation": flop,
                                        "ms_per_generation": stage["ref_ms"] / args.steps}
        out["setup_s"] = {"noise_table": t_noise}
        out["theta_abs_sum_after"] = theta_sum
        if world == 1 and not args.no_cpu_baseline:
            crumb("cpu baseline")
            out["cpu_baseline"] = cpu_baseline(noise.noise, theta0, ref, config.noise_stdev, args.tslimit, 18)
        print(json.dumps(out), flush=True)
    if transport is not None:
        dist.destroy_process_group()
    engine.close()
    crumb("done")


if __name__ == "__main__":
    main()
