"""CPU: the tracked evidence is self-consistent -- the bench line committed under profiles/ quotes exactly the traffic regime and
counter fraction that bench.py derives TODAY from the committed PMC summary (the driver line said 0.16, the tracked
profile 0.415 for the same launch mix, because the profile predated the final PMC file)."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _latest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


def test_bench_profile_quotes_the_committed_pmc_summary():
    import bench
    path = _latest("r0[4-9]_bench_steps20_warmup5.json")
    if path is None:
        pytest.skip("no round >= 4 bench profile committed yet")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    r = d["roofline"]
    rnd = os.path.basename(path)[:3]
    kind = 5 if "k_fc_ring" in r["kernel"] else 3                                    # dne_profile.fc_full_kind: k_fc_ring (round 5) / k_fc_duo
    per_unit, src, regime = bench._pmc_traffic(kind, r["units_per_launch"])
    assert src == os.path.join("profiles", "%s_pmc.json" % rnd), (src, rnd)          # the same round's counters, not an older file
    assert regime == r["traffic_regime"] and regime.startswith("bench_mix")          # measured on the bench's own launch mix
    assert r["traffic"] == pytest.approx(per_unit * r["units_per_launch"], rel=1e-9)
    want = per_unit * r["units_per_launch"] / (r["avg_launch_ms"] * 1e-3) / bench.HBM_PEAK
    assert r["frac_counter"] == pytest.approx(want, rel=1e-9)
    alg = r["units_per_launch"] * bench.ALG_BYTES_PER_ENV_STEP / (r["avg_launch_ms"] * 1e-3) / bench.HBM_PEAK
    if "frac_basis" in r:   # round 6: the headline fraction is the counted one over the union of the concurrent launches
        assert r["frac"] == pytest.approx(r["concurrent_launches"]["frac_counter"], rel=1e-9) and r["frac"] < 1.0
        assert r["achieved"] == pytest.approx(r["frac"] * bench.HBM_PEAK / 1e9, rel=1e-9)
        assert r["frac_algorithmic"] == pytest.approx(alg, rel=1e-9)
        assert r["frac_algorithmic"] <= 1.0 or r.get("algorithmic_denominator_exceeds_peak") is True
    else:
        assert r["frac"] == pytest.approx(alg, rel=1e-9)
    f = r["floors"]
    assert r["bound"] == "valu+hbm" and f["source"] == src and f["frac_of_binding_floor"] == pytest.approx(
        max(f["hbm_distinct_rows_ms"], f["valu_issue_ms"]) / r["avg_launch_ms"], rel=1e-9)
    # the PMC summary's bench_mix regime was collected on a run whose dispatch count matched the bench's own launch count
    pmc = json.load(open(os.path.join(ROOT, src)))
    mix = [x for x in pmc["regimes"] if x["regime"].startswith("bench_mix")][0]
    assert mix["dispatches_match_bench"] and mix["hbm_bytes_per_unit"] == pytest.approx(per_unit)
    assert r.get("traffic_bytes_per_unit", per_unit) == pytest.approx(per_unit)
    if "frac_basis" not in r and r["frac"] > 1.0:                                   # an algorithmic fraction above 1 carries its flag
        assert r.get("denominator_exceeds_peak") is True and "frac_pair_sharing" in r["whole_job"]
    # cpu baseline: states the host it ran on and a best-of-sweep wall-clock rate
    c = d["cpu_baseline"]
    assert c["value"] == pytest.approx(max(x["rate_wall"] for x in c["sweep"])) and c["host"]["usable_cpus"] <= c["host"]["os_cpu_count"]
    assert c["cores"] in [x["workers"] for x in c["sweep"]] and c["gpu_over_cpu"] == pytest.approx(d["value"] / c["value"])


def test_design_quotes_the_profile():
    """DESIGN.md's results section carries the same headline figures as the tracked profile (one number each)"""
    path = _latest("r0[4-9]_bench_steps20_warmup5.json")
    if path is None:
        pytest.skip("no round >= 4 bench profile committed yet")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert ("%.3f M env-steps/s" % (d["value"] / 1e6)) in text, "%.3f M env-steps/s" % (d["value"] / 1e6)
    assert ("frac_counter** %.3f" % d["roofline"]["frac_counter"]) in text or ("`frac_counter` %.3f" % d["roofline"]["frac_counter"]) in text
