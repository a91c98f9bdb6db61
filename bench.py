#!/usr/bin/env python3
"""bench.py -- env-steps/sec/generation of the ES hot path (BASELINE.json metric) on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W

N > 1 needs no wrapper: this process then launches the N ranks itself, one per device (RANK / LOCAL_RANK / WORLD_SIZE in their
environment, the RCCL id and the ranks' agreement on the carrier travelling through pipes it owns) -- the launcher side of
gpu_implementation/neuroevolution/concurrent_worker.py:129-142, which starts one worker per visible device.  Under an
external launcher (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N) the ranks it started are used as
they are and meet through a private directory keyed by the launcher's pid and MASTER_PORT.

A "step" is one ES generation of the workload named in config.workload: sample the noise indices, evaluate
every antithetic pair of the population (reset, virtual-batch-norm reference pass, lock-step act -> env.step
until each episode ends), exchange (noise_idx, return) records between GPUs, centered-rank -> weighted noise
sum -> Adam on every rank.  The population (pop 5000 = 2500 pairs) is sharded round-robin across the N GPUs
("strong" scaling: total work is fixed, as the metric "Frostbite ES pop 5000 at 1/2/4/8 GPUs" states).
Inputs (noise table, theta, reference batch) are resident in HBM before the timed region.
value = sum of episode lengths over all ranks and timed generations / max-over-ranks wall time (es.py:332,341).

After the headline region the same command times BASELINE.json's other configurations (tools/workloads.py: Deep GA on both
networks, the NS-ES meta-population loop, the six-game loop, the 2-worker CPU reference path) and reports them under "extra",
each with its own roofline / CPU baseline.  Default: all of them at N = 1; at N > 1 the sharded ones (ga, nses, sweep: BASELINE
configs 4 and 5 are defined on 4 and 8 GPUs), run over the same ranks on the headline engine's communicator.

The ranks never import torch on the RCCL path: the engine is reached through ctypes, device synchronisation is
hipDeviceSynchronize inside the C ABI, and the exchange, the barrier and the max / sum over ranks are RCCL calls behind
dne_comm_*.  So the HIP runtime in the process is the one libdne_hip.so was built against (/opt/rocm), the same one the GPU
tests run on.  If the ranks cannot build an RCCL communicator the launch FAILS (exit 3) -- unless --allow-gloo-fallback is given,
in which case they agree, all of them and explicitly, on carrying the same 32-byte records over gloo; the JSON line says which
carrier ran ("comm").

Environment: ALE and ROMs do not exist in this image, so the emulator under wrap_deepmind is the
Frostbite-shaped SynthAtari fixture -- stated in "data".
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

ALG_BYTES_PER_ENV_STEP = 4 * 1009058 + 28224   # all member weights once + the u8 observation stack
HBM_PEAK = 8.0e12                              # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK = 157.3e12                       # MI355X_MICROARCH.md: dense fp32 MFMA
# reference pass (virtual batch norm): flops of one reference frame through one member's network (2 * MACs)
REF_FLOP_PER_FRAME = 2 * (441 * 256 * 16 + 121 * 256 * 32 + 3872 * 256)
FC_KERNELS = {   # the kernel behind the profiled ("full") fc launches of one evaluation (dne_profile.fc_full_kind)
    5: "dne::k_fc_ring<true, 8> (table-ordered streaming fc, round 5: eight (pair, k-slice) units that follow each other in the noise table per "
       "workgroup, one per wave, walking ONE table timeline; the table window they share lives in an LDS ring that a ninth, loader wave fills "
       "by LDS-DMA five ticks ahead -- since round 6 from a copy of the table scaled once per sigma, fl(sigma * eps) entry by entry -- every noise row passes "
       "the CU's vector-memory path once per workgroup instead of once per unit; base rows "
       "from a column-permuted copy of the fc matrix (16 bytes per lane and row), activations relu(bn2(y2)) left by k_conv12 and read back "
       "as LDS broadcasts; one workgroup per CU; every window of a lock-step with >= 1500 active pairs on the rank; bn3 + output layer + "
       "argmax follow in k_out, outside the timed bracket)",
    4: "dne::k_fc_sub<2, true, true> (the mid range's sub-slice fc: one wave per 128 / 120-row chain of a pair, whole rows per load, eight rows "
       "per stream in flight, no LDS, no barrier; the head folds the 32 chain sums -- a rank whose share starts below 451 pairs, e.g. N = 8)",
    3: "dne::k_fc_duo<2, true, true, 8, true> (table-ordered streaming fc: a wave takes two (pair, k-slice) units that are neighbours in the noise "
       "table and walks them in table lock-step, 8 rows per stream in flight; the four waves of a workgroup -- eight adjacent units -- "
       "walk one table timeline, an s_barrier per row block, so rows several units need reach HBM once; one workgroup per CU by register "
       "footprint; every window of a lock-step with >= 451 active pairs on the rank; bn3 + output layer + argmax follow in k_out, outside the "
       "timed bracket)",
    2: "dne::k_fc2<true, 4> (streaming fc + bn + out + argmax, two antithetic pairs per work item: every lock-step with "
       ">= 800 active pairs on the rank)",
    1: "dne::k_fc<2, false, true, 4> (streaming fc + bn + out + argmax, one pair per work item: every window with > 96 active "
       "pairs; the rank's share is too small for k_fc2)",
}
PMC_PROFILES = tuple(os.path.join("profiles", "r0%d_pmc.json" % r) for r in (6, 5, 4, 3, 2, 1))
EXTRAS = ("ga", "ga_large", "nses", "sweep", "config1", "predicted")
EXTRAS_MULTI = ("ga", "nses", "sweep")         # default at N > 1: BASELINE configs 4 / 5 are DEFINED on 4 / 8 GPUs (ga_large, config1: one rank)
SIMDS, SHADER_HZ = 1024, 2.4e9                 # MI355X: 256 CUs x 4 SIMDs; nominal shader clock

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
    timed on a bounded sample of generation 0 of the same workload (tools/workloads.py:cpu_es, which imports the oracle).
    Checker only -- never the product path."""
    import workloads
    return workloads.cpu_es(noise, theta, ref, sigma, tslimit, n_actions)


FC_KERNEL_TAG = {5: "k_fc_ring", 4: "k_fc_sub", 3: "k_fc_duo", 2: "k_fc2", 1: "k_fc<"}


def _pmc_profile(kind):
    """the newest committed PMC summary (profiles/rNN_pmc.json) that measured this run's streaming kernel, or (None, None)"""
    for rel in PMC_PROFILES:
        p = os.path.join(ROOT, rel)
        if not os.path.exists(p):
            continue
        try:
            d = json.load(open(p))
        except Exception:
            continue
        tag = FC_KERNEL_TAG.get(kind, "?")
        if any(tag in r.get("kernel", "") for r in d.get("regimes", [])) or tag in d.get("k_fc_step", {}).get("kernel", ""):
            return d, rel
    return None, None


def _pmc_traffic(kind, units_per_launch):
    """HBM-side bytes per env-step of the streaming fc kernel from the committed rocprofv3 PMC passes (FETCH_SIZE doubled for the
    16-byte streaming loads as MI355X_MICROARCH.md prescribes, plus WRITE_SIZE; separate --pmc runs).  Preferred: the regime
    `bench_mix` -- the counters summed over EVERY dispatch of the kernel in a run of this very command (tools/collect_pmc_bench_mix.sh:
    bench.py's own launch mix, divided by the units those launches processed).  Otherwise the fixed-width regime (tools/kbench.py)
    whose units per launch is closest to this run's.  A property of the kernel measured under the profiler, NOT a measurement of
    this run: bench.py cannot read hardware counters in-process.  (None, None, None) if no profile is committed."""
    d, rel = _pmc_profile(kind)
    if d is None:
        return None, None, None
    tag = FC_KERNEL_TAG.get(kind, "?")
    regimes = [r for r in d.get("regimes", []) if tag in r["kernel"]]
    mix = [r for r in regimes if r["regime"].startswith("bench_mix")]
    if mix:
        return float(mix[0]["hbm_bytes_per_unit"]), rel, mix[0]["regime"]
    if regimes:
        best = min(regimes, key=lambda r: abs(np.log(max(r["units_per_launch"], 1.0) / max(units_per_launch, 1.0))))
        return float(best["hbm_bytes_per_unit"]), rel, best.get("regime")
    k = d["k_fc_step"]
    return float(k["hbm_bytes_per_unit"]), rel, "one full-width window (DNE_NSUB=1, 2500 pairs)"


def _floors(kind, units_per_launch, avg_ms):
    """What really bounds the streaming kernel: it is neither at the algorithmic-bytes roofline nor anywhere near
    it by choice -- pairs and table neighbours share rows.  Two floors per launch, from the committed profile: every DISTINCT
    table row under the launch's slices once at 8 TB/s, and the kernel's own VALU issue time (SQ_ACTIVE_INST_VALU quad-cycles x 4
    over 1024 SIMDs at the nominal clock: what its instruction stream costs with every wave always ready)."""
    d, rel = _pmc_profile(kind)
    if d is None:
        return None
    tag = FC_KERNEL_TAG.get(kind, "?")
    regimes = [r for r in d.get("regimes", []) if tag in r["kernel"] and "floor_unique_rows_bytes_per_unit" in r]
    mix = [r for r in regimes if r["regime"].startswith("bench_mix")]
    r = mix[0] if mix else (min(regimes, key=lambda r: abs(np.log(max(r["units_per_launch"], 1.0) / max(units_per_launch, 1.0)))) if regimes else None)
    out = {"source": rel}
    if r is not None:
        out["hbm_distinct_rows_bytes_per_unit"] = r["floor_unique_rows_bytes_per_unit"]
        out["hbm_distinct_rows_ms"] = r["floor_unique_rows_bytes_per_unit"] * units_per_launch / HBM_PEAK * 1e3
    sq = d.get("sq", {}).get(tag)
    if sq:
        out["valu_insts_per_unit"] = sq["SQ_INSTS_VALU_per_unit"]
        out["valu_issue_ms"] = sq["SQ_ACTIVE_INST_VALU_per_unit"] * 4.0 * units_per_launch / SIMDS / SHADER_HZ * 1e3
        out["valu_source"] = sq.get("regime")
        for k in ("wave_cycles_split", "waves_per_simd"):
            if k in sq:
                out[k] = sq[k]
    bind = max(out.get("hbm_distinct_rows_ms", 0.0), out.get("valu_issue_ms", 0.0))
    if bind > 0:
        out["binding"] = "valu" if out.get("valu_issue_ms", 0.0) >= out.get("hbm_distinct_rows_ms", 0.0) else "hbm"
        out["frac_of_binding_floor"] = bind / avg_ms
    return out


# ------------------------------------------------------------------------------------------------ rendezvous of the ranks
class PipeRendezvous:
    """Self-launched ranks: one line of JSON per message over the pair of pipes the launcher handed to this rank."""

    def __init__(self, fds):
        r, w = (int(x) for x in fds.split(","))
        self.r, self.w = os.fdopen(r, "r"), os.fdopen(w, "w")

    def _send(self, obj):
        self.w.write(json.dumps(obj) + "\n"); self.w.flush()

    def _recv(self):
        line = self.r.readline()
        if not line:
            raise SystemExit("rank %d: the launcher closed the control pipe" % _RANK)
        return json.loads(line)

    def exchange_uid(self, rank, uid, err):
        if rank == 0:
            self._send({"uid": uid.hex() if uid else None, "err": err})
        m = self._recv()
        return (bytes.fromhex(m["uid"]) if m.get("uid") else None), m.get("err")

    def vote(self, rank, world, ok, err):
        self._send({"ok": bool(ok), "err": err})
        return self._recv()

    def done(self, rank):
        pass


class FileRendezvous:
    """Ranks of an external launcher (torch.distributed.run): files in a directory only they can name -- the launcher's pid,
    MASTER_PORT and its run id -- and only this user can read.  Files older than this launch are ignored."""

    def __init__(self, world):
        # one directory per LAUNCH: the launcher's pid alone can recur, its start time (field 22 of /proc/<pid>/stat) cannot, and an
        # elastic restart inside one launcher bumps TORCHELASTIC_RESTART_COUNT -- a crashed launch's files are never in this one's way
        ppid = os.getppid()
        try:
            started = open("/proc/%d/stat" % ppid).read().rsplit(")", 1)[1].split()[19]
        except (OSError, IndexError):
            started = "0"
        key = "%s.%s.%s.%d.%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                  os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), ppid, started)
        self.dir = os.environ.get("DNE_RDV_DIR") or os.path.join("/tmp", "dne_rdv.%d.%s" % (os.getuid(), key))
        os.makedirs(self.dir, mode=0o700, exist_ok=True)
        self.fresh = _T0 - 300
        self.token = None           # rank 0's nonce, published with the id and echoed by every vote: a vote without it is not of this launch

    def _put(self, name, obj):
        p = os.path.join(self.dir, name)
        with open(p + ".tmp", "w") as f:
            json.dump(obj, f)
        os.replace(p + ".tmp", p)

    def _get(self, name, timeout, what, token=None):
        p, deadline = os.path.join(self.dir, name), time.time() + timeout
        while True:
            try:
                if os.path.getmtime(p) >= self.fresh:
                    m = json.load(open(p))
                    if token is None or m.get("token") == token:
                        return m
            except (OSError, ValueError):
                pass
            if time.time() > deadline:
                return None
            time.sleep(0.01)

    def exchange_uid(self, rank, uid, err):
        if rank == 0:
            for f in os.listdir(self.dir):               # leftovers of a crashed launch with the same key
                try:
                    os.unlink(os.path.join(self.dir, f))
                except OSError:
                    pass
            self._put("uid", {"uid": uid.hex() if uid else None, "err": err, "token": os.urandom(8).hex()})
        m = self._get("uid", 600, "the RCCL id")
        if m is None:
            return None, "rank %d: no RCCL id from rank 0 after 600 s" % rank
        self.token = m.get("token")
        return (bytes.fromhex(m["uid"]) if m.get("uid") else None), m.get("err")

    def vote(self, rank, world, ok, err):
        self._put("vote.%d" % rank, {"ok": bool(ok), "err": err, "token": self.token})
        votes = [self._get("vote.%d" % r, 400, "rank %d's vote" % r, token=self.token) for r in range(world)]
        errors = ["rank %d: %s" % (r, (v or {}).get("err") or "no answer") for r, v in enumerate(votes) if not (v and v["ok"])]
        return {"carrier": "gloo" if errors else "rccl", "errors": errors}

    def done(self, rank):
        if rank == 0:
            shutil.rmtree(self.dir, ignore_errors=True)


EXIT_NO_RCCL = 3


def gloo_fallback_refusal(decision, allowed):
    """--transport rccl (the default): a vote that ends on gloo is a FAILED launch unless --allow-gloo-fallback was given, so that
    no N > 1 run can report a scaling number that never touched RCCL.  Returns the message to leave with, or None to carry on."""
    if decision.get("carrier") == "rccl" or allowed:
        return None
    return ("no RCCL communicator on every rank (%s) and --allow-gloo-fallback not given: refusing to run the exchange over gloo "
            "(exit %d)" % ("; ".join(decision.get("errors") or ["no reason reported"]), EXIT_NO_RCCL))


def comm_init_bounded(engine, rank, world, uid, timeout):
    """ncclCommInitRank is a collective: if another rank never arrives it blocks.  It runs on its own host thread and this one
    waits `timeout` seconds; a rank that gives up votes against RCCL and every rank then takes the other carrier."""
    res = {}

    def run():
        try:
            engine.comm_init(rank, world, uid)
            res["ok"] = True
        except Exception as e:      # DneError with RCCL's own text
            res["err"] = str(e)
    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(timeout)
    if t.is_alive():
        return False, "ncclCommInitRank did not return within %d s" % timeout, True
    return bool(res.get("ok")), res.get("err"), False


# ------------------------------------------------------------------------------------------------ launchers
def supervise():
    """N = 1 only: run the benchmark proper in a child process.  A GPU memory-access fault is raised by the HSA runtime as an
    abort of the whole process -- no exception, no JSON (the fate of round 1's driver run).  The child's breadcrumbs name the
    stage it died in; one retry with the engine's own trace on (DNE_TRACE=1) then either yields a complete, separately timed
    run -- reported with "attempts": 2 and the first attempt's last stage -- or a second, more detailed failure."""
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, child_argv=None, check_devices=True):
    """--gpus N > 1 without an external launcher: start the N ranks (one per device, or all on device 0 with --single-device),
    carry the RCCL id from rank 0 to the others and the ranks' votes on the carrier back and forth, wait for them.  Rank 0
    prints the JSON line; every rank's breadcrumbs go to this process's stderr."""
    n = args.gpus
    need = 1 if args.single_device else n
    ndev = need
    if check_devices:
        from dne_hip import _lib
        try:
            ndev = _lib.device_count()
        except _lib.DneError as e:
            ndev = 0
            crumb("device count unavailable: %s" % e)
    if ndev < need:
        sys.stderr.write("bench.py --gpus %d needs %d HIP devices, this box shows %d.  RCCL takes one rank per device; to exercise the "
                         "%d-rank path on fewer devices run:  python bench.py --gpus %d --single-device --transport gloo\n"
                         % (n, need, ndev, n, n))
        return 2
    if args.single_device and args.transport == "rccl":
        sys.stderr.write("bench.py: --single-device puts every rank on device 0, which RCCL refuses; add --transport gloo\n")
        return 2
    port = _free_port()
    procs, to_child, from_child = [], [], []
    for r in range(n):
        p2c_r, p2c_w = os.pipe()
        c2p_r, c2p_w = os.pipe()
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DNE_BENCH_CHILD="1", DNE_CTRL_FDS="%d,%d" % (p2c_r, c2p_w), DNE_LAUNCHER="self")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: what RCCL's peer-to-peer setup needs on this driver
        procs.append(subprocess.Popen(child_argv or [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      pass_fds=(p2c_r, c2p_w), stdout=None if r == 0 else subprocess.DEVNULL))
        os.close(p2c_r); os.close(c2p_w)
        to_child.append(os.fdopen(p2c_w, "w")); from_child.append(os.fdopen(c2p_r, "r"))

    def recv(r):
        try:
            line = from_child[r].readline()
            return json.loads(line) if line else None
        except (OSError, ValueError):
            return None

    def send_all(obj):
        for w in to_child:
            try:
                w.write(json.dumps(obj) + "\n"); w.flush()
            except (OSError, ValueError):
                pass

    def control():
        if args.transport != "rccl":
            return
        m = recv(0) or {"uid": None, "err": "rank 0 ended before it produced an RCCL id"}
        send_all(m)
        votes = [recv(r) for r in range(n)]
        errors = ["rank %d: %s" % (r, (v or {}).get("err") or "ended without voting") for r, v in enumerate(votes) if not (v and v.get("ok"))]
        send_all({"carrier": "gloo" if errors else "rccl", "errors": errors})
    ct = threading.Thread(target=control, daemon=True)
    ct.start()
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
                sys.stderr.write("[bench launcher] rank %d ended with rc %d: stopping the other ranks\n" % (r, code))
                time.sleep(2.0)
                for o in live:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


# ------------------------------------------------------------------------------------------------ the extra workloads
def run_extras(which, noise, engine, rank, world, local_rank, transport_bytes, transport_records, tslimit, want_cpu, small, cpu_ref=None,
               steps=2, warmup=1, pop=5000, headline=None):
    """BASELINE configs 1, 3, 4, 5 (tools/workloads.py), each timed on its own; a failure of one is reported in its slot.
    small (--extra-small, the tests): the same code paths at a population of 96 / 48 children / two games."""
    import workloads as W
    out = {}
    shared = dict(device_id=local_rank, rank=rank, world=world, comm_from=engine if transport_bytes is None else None, tslimit=tslimit)
    ga_kw = dict(children=48, parents=6, generations=2) if small else {}
    ns_kw = dict(pop=96, archive_extra=2, iterations=1) if small else {}
    sw_kw = dict(pop=96, games=["frostbite", "asteroids"]) if small else {}

    def leg(name, fn):
        if name not in which and not (name.startswith("predicted_n") and "predicted" in which):
            return
        crumb("extra: %s" % name)
        t0 = time.time()
        try:
            out[name] = fn()
            out[name]["bench_wall_s"] = time.time() - t0
        except Exception as e:     # an extra never costs the headline line
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    leg("ga", lambda: W.ga_small(noise, transport=transport_bytes, **shared, **ga_kw))
    if world == 1:
        leg("ga_large", lambda: W.ga_large(noise, device_id=local_rank, tslimit=tslimit, **ga_kw))
    leg("nses", lambda: W.nses(noise, transport=transport_bytes, **shared, **ns_kw))
    leg("sweep", lambda: W.six_games(noise, transport=transport_records, **shared, **sw_kw))
    if world == 1 and "predicted" in which:
        # an N = 2 / 4 / 8 launch rehearsed on this one GPU, same generations as the headline region
        # (tools/workloads.py:simulate_ranks); shares = what ONE rank of such a launch spends per generation, beside the headline's own
        shares = {}
        if headline:
            shares["pairs_%d" % (pop // 2)] = {"ms_per_generation": round(headline["ms_per_step"], 2), "env_steps_per_s": round(headline["value"]), "source": "the headline region"}
        for n in ((2,) if small else (2, 4, 8)):
            leg("predicted_n%d" % n, lambda n=n: W.simulate_ranks(noise, n, steps=steps, warmup=warmup, pop=pop, tslimit=tslimit, device_id=local_rank,
                                                                   verify_generations=1 if small else 2))
            r = out.get("predicted_n%d" % n, {})
            if "error" not in r and r:
                shares["pairs_%d" % r["pairs_per_rank"]] = {"ms_per_generation": r["rank0_share_ms_per_generation"], "ms_per_generation_slowest_rank_plus_update": round(r["ms_per_step"], 2),
                                                            "source": "extra.predicted_n%d (rank 0's shard: evaluation + records_pack)" % n}
                if headline:
                    r["predicted_speedup_over_this_run"] = r["value"] / headline["value"]
        out["shares"] = shares
    ns_in = out.get("nses", {}).pop("_cpu_inputs", None)      # arrays for the CPU leg, not part of the report
    # the legs that run the ES network stream through the same kernel as the headline: price their whole-job rate at the bytes per
    # member-step the committed PMC summary holds for it (bench_mix regime) -- the streaming kernel's counted traffic, not the whole job's
    per_unit, src, regime = _pmc_traffic(5, 1000.0)
    if per_unit is None:
        per_unit, src, regime = _pmc_traffic(3, 1000.0)
    for name in ("nses", "sweep"):
        r = out.get(name, {}).get("roofline")
        if r and per_unit:
            # one unit for `traffic` in every roofline object of the line: counted bytes per member-step
            r["traffic"] = None
            r["traffic_bytes_per_unit"] = per_unit
            r["traffic_GBps"] = out[name]["value"] * per_unit / 1e9
            r["traffic_source"] = "counted bytes of the ES streaming kernel at this leg's env-steps/s (%s, %s)" % (src, (regime or "").split(":")[0])
            r["frac_counter"] = out[name]["value"] * per_unit / (HBM_PEAK * world)
    # the GA legs: every dispatch of tools/ga_bench.py (--large) counted, per env-step of that run (tools/summarize_pmc_ga.py)
    for rel in [r for r in ("profiles/r06_pmc_ga.json", "profiles/r05_pmc_ga.json") if os.path.exists(os.path.join(ROOT, r))][:1]:
        try:
            gd = json.load(open(os.path.join(ROOT, rel)))
        except Exception:
            continue
        for name in ("ga", "ga_large"):
            r = out.get(name, {}).get("roofline") if isinstance(out.get(name), dict) else None
            if r and name in gd:
                r["traffic"] = None
                r["traffic_bytes_per_unit"] = gd[name]["bytes_per_unit"]
                r["traffic_GBps"] = out[name]["value"] * gd[name]["bytes_per_unit"] / 1e9
                r["traffic_source"] = "%s: FETCH_SIZE x2 + WRITE_SIZE over every dispatch of tools/ga_bench.py%s, per env-step of that run" % (rel, " --large" if name == "ga_large" else "")
                r["frac_counter"] = out[name]["value"] * gd[name]["bytes_per_unit"] / (HBM_PEAK * world)
    for name in out:   # an algorithmic fraction above 1 says the denominator counts bytes the kernels share, not that a roofline was beaten
        r = out[name].get("roofline") if isinstance(out[name], dict) else None
        if r and isinstance(r.get("frac"), (int, float)) and r["frac"] > 1.0:
            r["denominator_exceeds_peak"] = True
    if rank == 0 and want_cpu:
        # the CPU legs of the extras run at the worker count the headline sweep found best (cpu_ref), on bounded samples
        procs = (cpu_ref or {}).get("cores") or None
        if "ga" in out and "error" not in out["ga"]:
            crumb("extra: ga cpu baseline")
            out["ga"]["cpu_baseline"] = W.cpu_ga(noise.noise, 0.005, tslimit, 18, children=ga_kw.get("children", 1000), procs=procs)
        if ns_in is not None and "error" not in out["nses"]:
            crumb("extra: nses cpu baseline")
            out["nses"]["cpu_baseline"] = W.cpu_nses(noise.noise, ns_in["theta"], ns_in["ref"], ns_in["archive"], ns_in["k"], 0.02, tslimit, 18,
                                                     n_pairs_total=ns_kw.get("pop", 5000) // 2, procs=procs, sample_pairs=4 if small else None)
        if "sweep" in out and "error" not in out["sweep"]:
            crumb("extra: sweep cpu baseline")
            out["sweep"]["cpu_baseline"] = W.cpu_sweep(noise.noise, out["sweep"]["games"], cpu_ref, tslimit, procs, sample_pairs=4 if small else None)
        leg("config1", lambda: W.config1_cpu(noise.noise, tslimit=tslimit, sample_pairs=2 if small else 8))
    return out


def _gloo_allgather_bytes(buf, world):
    """host all-gather of one numpy record array per rank (the GA / NS-ES drivers' `transport`)"""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(buf).view(np.uint8).reshape(-1).copy())
    out = torch.empty(world * t.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(out, t)
    return out.numpy().view(buf.dtype)


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.single_device:
        local_rank = 0
    launcher = os.environ.get("DNE_LAUNCHER") or ("external (RANK / WORLD_SIZE found in the environment)" if world > 1 else "none")
    crumb("start: world %d, steps %d, warmup %d" % (world, args.steps, args.warmup))
    from dne_hip import _lib, es, policies   # loads libdne_hip.so (and with it /opt/rocm's HIP runtime) first
    n_pairs = args.pop // 2
    config = es.Config(**EXP["config"])
    my_pairs = len(es.shard_pairs(n_pairs, rank, world))
    try:
        try:
            ndev = _lib.device_count()
        except AttributeError:      # DNE_LIB_PATH points at a build that predates dne_device_count
            ndev = local_rank + 1
        if local_rank >= max(ndev, 1):
            raise _lib.DneError("rank %d wants device %d, this box shows %d" % (rank, local_rank, ndev))
        engine = _lib.Engine(_lib.KIND_ES, 18, max_members=2 * my_pairs, ref_count=128, device_id=local_rank,
                             profile_events=not args.no_profile_events)
    except _lib.DneError as e:
        raise SystemExit("bench.py needs %d MI355X device(s): the HIP engine has no CPU fallback (%s)" % (1 if args.single_device else world, e))
    crumb("engine created on device %d (%d pairs)" % (local_rank, my_pairs))
    comm = {"carrier": "none (one rank)", "launcher": launcher}
    transport = transport_bytes = None
    hung_init = False
    rdv = None
    if world > 1:
        rdv = PipeRendezvous(os.environ["DNE_CTRL_FDS"]) if os.environ.get("DNE_CTRL_FDS") else FileRendezvous(world)
        use_gloo = args.transport == "gloo"
        if not use_gloo:
            uid = err = None
            if rank == 0:
                try:
                    uid = _lib.comm_unique_id()
                except _lib.DneError as e:
                    err = "rank 0 could not open RCCL: %s" % e
            uid, err = rdv.exchange_uid(rank, uid, err)
            ok = False
            if uid is not None:
                ok, err, hung_init = comm_init_bounded(engine, rank, world, uid, args.rccl_init_timeout)
            decision = rdv.vote(rank, world, ok, err)
            refusal = gloo_fallback_refusal(decision, args.allow_gloo_fallback)
            if refusal:
                # the default carrier is RCCL and a scaling run must not "pass" on the host path: every rank got the same decision
                # and every rank leaves with the same code
                crumb(refusal)
                engine.comm_abort()
                if rdv is not None:
                    rdv.done(rank)
                sys.stdout.flush(); sys.stderr.flush()
                os._exit(EXIT_NO_RCCL)      # a thread may still sit inside ncclCommInitRank
            if decision["carrier"] == "rccl":
                engine.barrier()
                r_, n_, is_ = engine.comm_info()
                if n_ != world or r_ != rank:
                    raise SystemExit("rank %d: the RCCL communicator reports ncclCommCount %d / ncclCommUserRank %d, the launch has %d ranks"
                                     % (rank, n_, r_, world))
                comm.update({"carrier": "rccl", "nccl_comm_count": n_, "nccl_user_rank": r_})
                crumb("RCCL communicator ready: ncclCommCount %d, ncclCommUserRank %d" % (n_, r_))
            else:
                # every rank takes this branch together (the launcher, or the vote files, handed all of them the same decision).
                # The records are 32 bytes per pair: the carrier decides nothing about the result, only ~1 ms of latency.
                engine.comm_abort()
                use_gloo = True
                comm["rccl_error"] = decision["errors"]
                crumb("no RCCL communicator (%s): the 32-byte records travel over gloo" % "; ".join(decision["errors"]))
        if use_gloo:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            transport, transport_bytes = es.allgather_records, _gloo_allgather_bytes
            comm["carrier"] = "gloo"
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
    all_units = all_launches = 0      # streaming-kernel units / launches of EVERY generation incl. warm-up: the denominator of a PMC pass
    for _ in range(args.warmup):
        crumb("generation %d (warmup) eval" % gen)
        es.es_generation(engine, noise.noise.size, config, n_pairs, gen, args.tslimit, EXP["optimizer"], rank, world, transport)
        p = engine.profile()
        all_units += p["fc_full_units"]; all_launches += p["fc_full_launches"]
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
        all_units += p["fc_full_units"]; all_launches += p["fc_full_launches"]
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

    which = [x for x in (args.extra.split(",") if args.extra not in ("", "none", "all", "default") else
                         (EXTRAS if args.extra == "all" or (args.extra == "default" and world == 1) else
                          EXTRAS_MULTI if args.extra == "default" else ())) if x]
    for x in which:
        if x not in EXTRAS:
            raise SystemExit("--extra: unknown workload %r (choose from %s)" % (x, ", ".join(EXTRAS)))
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # before the extras: their CPU legs reuse the worker count it finds
        crumb("cpu baseline")
        cpu = cpu_baseline(noise.noise, theta0, ref, config.noise_stdev, args.tslimit, 18)
    extra = None
    if which:
        extra = run_extras(which, noise, engine, rank, world, local_rank, transport_bytes, transport, args.tslimit,
                           not args.no_cpu_baseline and world == 1, args.extra_small, cpu_ref=cpu,   # CPU legs: rank 0 at N = 1 only
                           steps=args.steps, warmup=args.warmup, pop=96 if args.extra_small else args.pop,
                           headline={"value": total_steps / wall, "ms_per_step": 1e3 * wall / max(args.steps, 1)})

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
                       "pairs_per_gpu": my_pairs,
                       "parallelism": "population sharded round-robin over %d GPU(s); all-gather of 32-byte records over %s; "
                                      "redundant update on every rank" % (world, comm["carrier"])},
            "comm": comm,
        }
        if fc_ms > 0:
            avg_ms = fc_ms / fc_launches
            units_per_launch = fc_units / fc_launches
            per_unit, src, regime = _pmc_traffic(fc_kind, units_per_launch)
            achieved = units_per_launch * ALG_BYTES_PER_ENV_STEP / (avg_ms * 1e-3)
            traffic = per_unit * units_per_launch if per_unit else None
            floors = _floors(fc_kind, units_per_launch, avg_ms)
            out["roofline"] = {
                "bound": "valu+hbm" if floors and "valu_issue_ms" in floors else "hbm", "kernel": FC_KERNELS[fc_kind],
                "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
                "frac_algorithmic": achieved / HBM_PEAK,
                "frac_counter": (traffic / (avg_ms * 1e-3) / HBM_PEAK) if traffic else None,
                "traffic": traffic, "traffic_bytes_per_unit": per_unit, "traffic_regime": regime,
                "traffic_source": ("%s: PMC bytes per env-step of this kernel from separate rocprofv3 --pmc passes (FETCH_SIZE x2 + "
                                   "WRITE_SIZE) at the regime named in traffic_regime, scaled to this run's units per launch -- not "
                                   "measured in this run" % src) if src else None,
                "algorithmic_bytes_per_unit": ALG_BYTES_PER_ENV_STEP, "unit_def": "one env-step of one member",
                "units_per_launch": units_per_launch, "avg_launch_ms": avg_ms, "launches": int(fc_launches),
                "all_generations": {"units": int(all_units), "launches": int(all_launches)},
                "floors": floors,
                "note": "frac / achieved = COUNTED bytes (traffic_bytes_per_unit x units) over the union of the concurrent launches; "
                        "frac_algorithmic / achieved_algorithmic = SURVEY 8d bytes (every member's weights once per env-step) / average launch "
                        "time / 8 TB/s -- above the peak because an antithetic pair shares one read of its noise slice and (k_fc_ring / k_fc_duo) "
                        "neighbouring units share table rows: an invalid denominator, not skipped work (the bit-exact generation tests prove the "
                        "work); frac_counter = counted bytes per launch / average launch time; floors = what bounds the kernel now: its distinct "
                        "table rows once at 8 TB/s and its own VALU issue time (SQ counters), per launch",
            }
            # SURVEY 8d also asks for the whole-job figure: every env-step of the generation (reference pass, tail and
            # update included in the time) priced at the same algorithmic bytes
            PAIR_BYTES = 4 * 3872 * 256 / 2 + 28224   # 2 010 688: a pair's fc noise slice once for both members + the u8 stack (profiles/*_pmc.json: floor_pair_sharing_bytes_per_unit)
            out["roofline"]["whole_job"] = {"achieved": value * ALG_BYTES_PER_ENV_STEP / 1e9, "unit": "GB/s",
                                            "frac": value * ALG_BYTES_PER_ENV_STEP / (HBM_PEAK * world),
                                            "frac_pair_sharing": value * PAIR_BYTES / (HBM_PEAK * world),
                                            "pair_sharing_bytes_per_unit": PAIR_BYTES}
            # the windows launch this kernel concurrently from several streams; a launch that shares the chip with its siblings is
            # stretched, so also: all profiled bytes / the time during which at least one such launch was running
            if fc_union_ms > 0:
                uni = fc_units * ALG_BYTES_PER_ENV_STEP / (fc_union_ms * 1e-3)
                out["roofline"]["concurrent_launches"] = {"achieved": uni / 1e9, "unit": "GB/s", "frac": uni / HBM_PEAK,
                                                          "frac_counter": (uni / ALG_BYTES_PER_ENV_STEP * per_unit / HBM_PEAK) if per_unit else None,
                                                          "busy_ms_per_generation": fc_union_ms / args.steps}
            for o in (out["roofline"]["whole_job"], out["roofline"].get("concurrent_launches", {})):
                if o.get("frac", 0) > 1.0:   # the SURVEY 8d denominator counts shared bytes once per member: not a roofline beaten
                    o["denominator_exceeds_peak"] = True
            if out["roofline"]["frac_algorithmic"] > 1.0:
                out["roofline"]["algorithmic_denominator_exceeds_peak"] = True
            # The headline fraction: COUNTED bytes of the kernel (FETCH_SIZE x2 + WRITE_SIZE per unit from the
            # committed --pmc passes over this very command) x the units this run's launches processed / the time during which at least
            # one such launch was running (the windows launch it concurrently: per-launch durations overlap) / 8 TB/s.  The SURVEY 8d
            # figure stays beside it as frac_algorithmic / achieved_algorithmic with its flag.
            cl = out["roofline"].get("concurrent_launches", {})
            if cl.get("frac_counter") is not None:
                out["roofline"]["frac"] = cl["frac_counter"]
                out["roofline"]["frac_basis"] = "counted HBM bytes per unit x units / union of the kernel's concurrent launches / 8 TB/s"
            elif out["roofline"]["frac_counter"] is not None:
                out["roofline"]["frac"] = out["roofline"]["frac_counter"]
                out["roofline"]["frac_basis"] = "counted HBM bytes per unit x units per launch / average launch duration / 8 TB/s"
            else:
                out["roofline"]["frac_basis"] = "algorithmic bytes (no committed counter profile for this kernel)"
            out["roofline"]["achieved_algorithmic"] = out["roofline"]["achieved"]
            out["roofline"]["achieved"] = out["roofline"]["frac"] * HBM_PEAK / 1e9
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
        if extra is not None:
            out["extra"] = extra
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["cpu_baseline"]["gpu_over_cpu"] = value / cpu["value"]
            out["cpu_baseline"]["note"] = ("the GPU box grants this container %s CPUs of its %s (cgroup cpu.max / affinity: host); the ratio is against "
                                           "THAT host share -- per CPU-second the oracle makes %.0f env-steps, so a whole host of %s cores at the "
                                           "same per-core rate would be about %.0f env-steps/s (memory bandwidth permitting)"
                                           % (cpu["host"].get("usable_cpus"), cpu["host"].get("os_cpu_count"), cpu["rate_per_cpu_second"],
                                              cpu["host"].get("physical_cores"), cpu["rate_per_cpu_second"] * (cpu["host"].get("physical_cores") or 0)))
        print(json.dumps(out), flush=True)
    if transport is not None:
        dist.barrier()
        dist.destroy_process_group()
    elif world > 1:
        engine.barrier()
    if rdv is not None:
        rdv.done(rank)
    crumb("done")
    if hung_init:       # a thread of this process is still inside ncclCommInitRank: leave without waiting for it
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)
    engine.close()


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
    ap.add_argument("--extra", default="default",
                    help="BASELINE configs timed after the headline region: comma list of %s, or all / none; default = all at N = 1, "
                         "%s at N > 1 (configs 4 / 5 are defined on 4 / 8 GPUs)" % (", ".join(EXTRAS), ",".join(EXTRAS_MULTI)))
    ap.add_argument("--extra-small", action="store_true", help="testing aid: the extra workloads at a population of 96")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "gloo"],
                    help="exchange for N > 1: rccl = dne_comm_* (RCCL over xGMI behind the C ABI); gloo = torch.distributed on the "
                         "host, only to exercise the multi-rank path on a box with fewer GPUs than ranks")
    ap.add_argument("--single-device", action="store_true", help="testing aid: every rank uses GPU 0 (needs --transport gloo)")
    ap.add_argument("--allow-gloo-fallback", action="store_true",
                    help="--transport rccl: if any rank cannot build the RCCL communicator, carry the records over gloo instead of "
                         "exiting with code %d (the JSON line's comm.carrier then says gloo)" % EXIT_NO_RCCL)
    ap.add_argument("--rccl-init-timeout", type=int, default=int(os.environ.get("DNE_RCCL_INIT_TIMEOUT", "180")))
    ap.add_argument("--no-supervisor", action="store_true", help="N = 1: run in this process (no child, no retry)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    launched = "WORLD_SIZE" in os.environ or os.environ.get("DNE_BENCH_CHILD")
    if not launched:
        if args.gpus > 1:
            sys.exit(launch_ranks(args))
        if not args.no_supervisor:
            sys.exit(supervise())
    run_rank(args)


if __name__ == "__main__":
    main()
