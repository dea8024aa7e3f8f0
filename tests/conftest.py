import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    # CPU oracle forwards: a GPU box shows all 128 host cores but a job's share is 16, and torch at 128 threads runs the oracle
    # network 8x SLOWER than at 16 (bench.py cpu_baseline probe: 1.92 s vs 0.25 s per forward) -- one setting for every test
    import os
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. a plain `pytest tests` on CPU."""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rnd(seed, *shape):
    """Same seeded-input recipe as tests/golden/make_goldens.py."""
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape, dtype=np.float32))


def gamma_raw(count, k, theta, shape):
    """Draw number ``count`` of tests/golden/make_goldens.py::gen_sampler_gamma (same recipe as its ``gamma_raw``): the
    stand-in for a Gamma(k, rate 1/theta) sample, k theta + n / 2 with n = rnd(500 + count): IEEE fp32 mul / div / add only,
    so it is the same tensor on every host."""
    kk = torch.full(tuple(shape[1:]), float(k))
    th = 1.0 / torch.full(tuple(shape[1:]), float(1 / theta))
    return kk * th + 0.5 * rnd(500 + count, *shape)


def gamma_feed(g, name, first, k_cum, theta_t, steps, shape=(2, 15, 32, 32)):
    """noise_fn for a ``gamma=True`` run of golden ``name``: the generator's raw draws, numbered from ``first``, each
    checked against the checksum the reference run recorded.  Returns (noise_fn, number of draws)."""
    sums = g[name + "_raw_sums"]
    state = {"n": 0}
    ks, th = k_cum[steps], theta_t[steps]

    def fn(tag, x):
        i = tag if tag != "t_min" else state["first_step"]
        raw = gamma_raw(first + state["n"], ks[i], th[i], shape)
        assert float(raw.double().sum()) == float(sums[state["n"]]), (name, tag)
        state["n"] += 1
        return raw
    return fn, state, len(sums)
