"""CPU: tools/workloads.py (BASELINE configs 1, 3, 4, 5 as bench.py's `extra` block times them) driven through the oracle-backed
engine at toy sizes -- the loops, the record keeping and the roofline arithmetic, not the speed."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture()
def W(oracle, monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_engine import OracleEngine
    from dne_hip import _lib
    import workloads

    def make(kind, n_actions=18, max_members=64, ref_count=16, device_id=0, **kw):
        assert kind in (_lib.KIND_ES, _lib.KIND_GA)
        return OracleEngine(kind, n_actions, max_members, 128 if kind == _lib.KIND_ES else 0, **kw)
    monkeypatch.setattr(_lib, "Engine", make)
    return workloads


class Noise:
    def __init__(self, n=3_000_000):
        self.noise = np.random.RandomState(123).randn(n).astype(np.float32)
        self._engines = []

    def attach(self, e):
        e.noise_upload(self.noise)
        self._engines.append(e)

    def sample_index(self, stream, dim):
        return stream.randint(0, len(self.noise) - dim + 1)


@pytest.mark.timeout(900)
def test_ga_generations_and_roofline(W):
    r = W.ga_small(Noise(), generations=2, children=6, parents=3, tslimit=8)
    assert [g["gen"] for g in r["generations"]] == [0, 1] and all(g["env_steps"] > 0 and g["max_len"] <= 8 for g in r["generations"])
    assert r["value"] == pytest.approx(r["generations"][1]["steps_per_s"])          # generation 0 (root genomes) is not the value
    assert r["roofline"]["achieved"] == pytest.approx(r["value"] * (4 * 1008450 + 28224) / 1e9) and r["roofline"]["traffic"] is None
    # generations 10 / 100 and the chain of 259 -- synthetic deep genomes, cold and warm parent cache
    assert [d["chain"] for d in r["deep_chains"]] == [10, 100, 259]
    assert all(d["cold"]["env_steps"] == d["warm"]["env_steps"] > 0 for d in r["deep_chains"])


@pytest.mark.timeout(900)
def test_six_games_use_each_games_action_count(W):
    r = W.six_games(Noise(), generations=1, warmup=0, pop=4, tslimit=6, games=["frostbite", "asteroids"])
    assert [(g["game"], g["n_actions"], g["num_params"]) for g in r["games"]] == [("frostbite", 18, 1009058), ("asteroids", 14, 1009058 - 4 * 257)]
    assert r["value"] > 0 and r["roofline"]["frac"] == pytest.approx(r["value"] * 4064456 / 8e12)


@pytest.mark.timeout(900)
def test_cpu_legs_time_the_oracle_like_reference_workers(W, oracle):
    noise = Noise().noise
    r = W.config1_cpu(noise, pop=8, tslimit=6, sample_pairs=2)
    assert r["cores"] == 2 and r["kind"] == "port" and r["value"] > 0 and "2 CPU worker processes" in r["workload"]
    g = W.cpu_ga(noise, 0.005, 6, 18, children=4, procs=2, sample=2)
    assert g["cores"] == 2 and g["value"] > 0
    # the sweep: every leg reports wall AND consumed CPU seconds; the value is the best wall-clock rate; the host is described
    from dne_hip import policies
    th, ref = policies.xavier_flat(18, 0), oracle.get_ref_batch(seed=0, batch_size=4)
    s = W.cpu_es(noise, th, ref, 0.02, 6, 18, n_pairs_total=8, sample_pairs=2)
    import hostinfo
    u = min(hostinfo.usable_cpus(), os.cpu_count())     # one worker per granted CPU (launch.py:117), and twice that
    assert [r["workers"] for r in s["sweep"]] == [u, 2 * u] and all(r["cpu_s"] > 0 and r["wall_s"] > 0 for r in s["sweep"])
    assert s["value"] == max(r["rate_wall"] for r in s["sweep"]) and s["cores"] in [r["workers"] for r in s["sweep"]]
    h = s["host"]
    assert h["usable_cpus"] <= h["os_cpu_count"] and h["sched_getaffinity"] >= 1 and "cgroup_cpu_max" in h and h["model"]


def test_usable_cpus_honours_the_cgroup_quota(monkeypatch):
    import hostinfo
    monkeypatch.setattr(hostinfo, "affinity_cpus", lambda: 64)
    monkeypatch.setattr(hostinfo, "_read", lambda p: "1050000 100000" if p == "/sys/fs/cgroup/cpu.max" else None)
    assert hostinfo.cgroup_cpu_limit() == 10.5 and hostinfo.usable_cpus() == 11
    monkeypatch.setattr(hostinfo, "_read", lambda p: "max 100000" if p == "/sys/fs/cgroup/cpu.max" else None)
    assert hostinfo.cgroup_cpu_limit() is None and hostinfo.usable_cpus() == 64


@pytest.mark.timeout(900)
def test_nses_meta_population_iterations(W):
    r = W.nses(Noise(), iterations=2, pop=4, meta_pop=2, archive_extra=1, k=2, tslimit=6)
    its = r["iterations"]
    assert [i["archive"] for i in its] == [4, 5] and all(i["env_steps"] > 0 and np.isfinite(i["update_ratio"]) for i in its)
    assert r["value"] > 0 and "archive 3 -> 5" in r["workload"]
