import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "deep-neuroevolution_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")
    config.addinivalue_line("markers", "variants: a kernel / schedule no default path launches (the A/B library): -m 'gpu and variants' or DNE_TEST_VARIANTS=1")


def pytest_collection_modifyitems(config, items):
    """The default GPU suite covers the product path; the kernels kept only as same-box A/B references (forward_variants.h and the
    non-default schedules) run when asked for by name."""
    if os.environ.get("DNE_TEST_VARIANTS") == "1" or "variants" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="non-default kernel variant: run with -m 'gpu and variants' or DNE_TEST_VARIANTS=1")
    for it in items:
        if "variants" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))


@pytest.fixture(scope="session")
def small_noise():
    """First 4M entries of the reference noise stream (es.py:60, seed 123)."""
    return np.random.RandomState(123).randn(4_000_000).astype(np.float32)


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def noise_table():
    """The reference's 250M-entry table (es.py:51-61), sampled once per test session (2.5 s, 1 GB); engines attach() to it."""
    from dne_hip import es
    return es.SharedNoiseTable()


ES_FULL = dict(n_pairs=2500, nact=18, sigma=0.02, tslimit=5000)   # BASELINE.json configs[1]


@pytest.fixture(scope="session")
def oracle_es_gen0(oracle, noise_table):
    """Generation 0 of config 2 (= the rollouts of config 4's first iteration: same theta, reference batch, indices and seeds)
    on the CPU oracle over every usable host core, ONCE per session: all 2500 x 2 returns, sign-returns, lengths and RAM
    trajectories.  About a minute on the GPU box's host."""
    import oracle_pool
    from dne_hip import es, policies
    c = ES_FULL
    th = policies.xavier_flat(c["nact"], 0)
    ref = oracle.get_ref_batch(seed=0, batch_size=128, nact=c["nact"])
    _, idx, seeds = es.generation_inputs(noise_table.noise.size, th.size, c["n_pairs"], 0, 0, 1)
    ret, sg, ln, bcs = oracle_pool.es_generation(noise_table.noise, th, ref, idx, seeds, c["sigma"], c["tslimit"], c["nact"], want_bc=True)
    return dict(theta=th, ref=ref, idx=idx, seeds=seeds, ret=ret, sg=sg, ln=ln, bcs=bcs)
