"""GPU tier: the lifetime of device handles (round-4 verdict, item 7).  A handle used to own a HIP stream -- an HSA queue, with its scratch
arena -- and a process that kept many models alive (the reference's hyper-parameter loop, examples/exact_time_learning.jl, binds a model per
evaluation) aborted in the runtime with HSA_STATUS_ERROR_OUT_OF_RESOURCES at queue creation.  Handles now share a small per-device pool of
streams (tgp_api.hip `pool_stream`); this file builds thousands of models in one process WITHOUT any gc.collect() (the autouse fixture that
papered over it is gone from tests/conftest.py) and keeps hundreds alive at once."""
import numpy as np
import pytest

from oracle import components as oc
from oracle import lgssm_ref as ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tgp():
    import temporalgps_jl_amd as t
    t._lib.load()
    return t


def _model(tgp, spec, T, s2):
    m = oc.build_lgssm(spec, ("regular", 0.0, 0.1, T), s2)
    tr = tgp.GaussMarkovModel(tgp.Forward, m["A"], m["a"], m["Q"], tgp.Gaussian(m["x0m"], m["x0P"]))
    return m, tgp.LGSSM(tr, tgp.ScalarOutputLGC(m["H"], np.atleast_1d(m["h"]), np.atleast_1d(m["R"])), T=T)


def test_five_thousand_models_built_and_dropped(tgp):
    """a hyper-parameter loop: a new model per evaluation, the old one left to the reference count"""
    rng = np.random.default_rng(0)
    T = 300
    y = rng.standard_normal(T)
    specs = [("matern32",), ("matern52",), ("sum", ("matern32",), ("matern12",))]
    checked = 0
    for i in range(5000):
        s2 = 0.05 + 0.001 * (i % 100)
        m, dm = _model(tgp, specs[i % 3], T, s2)
        lp = tgp.logpdf(dm, y)
        if i % 500 == 0:
            want = ref.logpdf(m, y)
            assert abs(lp - want) <= 1e-10 * abs(want)
            checked += 1
    assert checked == 10


def test_six_hundred_models_alive_at_once(tgp):
    """... and a caller that keeps them all (a grid of fitted models): every one of them still answers"""
    rng = np.random.default_rng(1)
    T = 200
    y = rng.standard_normal(T)
    alive = []
    for i in range(600):
        m, dm = _model(tgp, ("matern52",), T, 0.05 + 0.001 * i)
        alive.append((m, dm, tgp.logpdf(dm, y)))
    for i in (0, 299, 599):
        m, dm, lp = alive[i]
        want = ref.logpdf(m, y)
        assert abs(lp - want) <= 1e-10 * abs(want)
        mean, var = tgp.posterior_marginals(dm, y, np.array([1e-18]))
        pm, pv = ref.marginals(ref.replace_observation_noise_cov(ref.posterior(m, y), np.full(T, 1e-18)))
        assert np.abs(mean - pm).max() <= 1e-8 and np.abs(var - pv).max() <= 1e-8


def test_handles_of_the_larger_state_dimensions_too(tgp):
    """the out-of-line kernels of d >= 9 bring a scratch arena per queue: two hundred such models alive"""
    rng = np.random.default_rng(2)
    from tests import _util as U
    T, d = 120, 10
    alive = []
    for i in range(200):
        model = U.random_lgssm(rng, False, d, T)
        tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
        dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=T)
        y = rng.standard_normal(T)
        lp = tgp.logpdf(dm, y)
        alive.append(dm)
        if i % 50 == 0:
            want = ref.logpdf(model, y)
            assert abs(lp - want) <= 1e-10 * abs(want)
