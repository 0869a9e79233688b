"""GPU tier: pass 2 of the lane-per-chunk smoother as two kernels (k_apply_filter<MODE 4> + k_compose_smoother;
tgp_chunk_body.inc: chunk_compose_smoother, TGP_OPT_SPLIT_SMOOTHER) against the fused MODE 2 kernel and the oracle
(lgssm.jl:193-238 posterior / invert_dynamics, :111-115 reverse marginals): d = 5 (forced), 6, 7 (default), shared and
per-step models, missing data, ragged chunks, scalar and vector observations.
Tolerances: split vs fused 1e-11 (same arithmetic, different kernels); vs the oracle as in test_gpu_parity.py."""
import numpy as np
import pytest

from oracle import lgssm_ref as ref
from tests import _util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tgp():
    import temporalgps_jl_amd as t
    t._lib.load()
    return t


def _names(hd, tgp, f):
    hd.set_option(tgp._lib.OPT_PROFILE, 1)
    hd.profile_reset()
    out = f()
    names = set(hd.profile())
    hd.set_option(tgp._lib.OPT_PROFILE, 0)
    return out, names


@pytest.mark.parametrize("d", [5, 6, 7])
@pytest.mark.parametrize("tv", [False, True])
def test_split_smoother_equals_fused_and_oracle(tgp, d, tv):
    rng = np.random.default_rng(31 * d + tv)
    T = 1777
    model = U.random_lgssm(rng, tv, d, T)
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    y = ref.rand(model, *eps)
    missing = rng.random(T) < 0.2
    ym = y.copy()
    ym[missing] = np.nan
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], model["h"], model["R"]), T=T)
    hd = dm.handle()
    hd.set_option(tgp._lib.OPT_GROUP, 0)                     # lane-per-chunk kernels throughout
    post = ref.posterior_missing(model, y, missing)
    Rn = rng.random(T) * 0.1
    pm, pC = ref.marginals(ref.replace_observation_noise_cov(post, Rn))
    for chunk in (0, 3, 13, 64):
        hd.set_option(tgp._lib.OPT_CHUNK, chunk)
        hd.set_option(tgp._lib.OPT_SPLIT_SMOOTHER, 0)
        (fm, fv), names = _names(hd, tgp, lambda: tgp.posterior_marginals(dm, ym, Rn))
        assert not any(n.startswith("k_compose_smoother") for n in names), names
        hd.set_option(tgp._lib.OPT_SPLIT_SMOOTHER, 2)
        (sm, sv), names = _names(hd, tgp, lambda: tgp.posterior_marginals(dm, ym, Rn))
        assert any(n.startswith("k_compose_smoother") for n in names), names
        np.testing.assert_allclose(sm, fm, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(sv, fv, rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(sm, pm, rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(sv, pC, rtol=1e-8, atol=1e-9)
    hd.set_option(tgp._lib.OPT_SPLIT_SMOOTHER, 1)           # default: split from d = 6
    _, names = _names(hd, tgp, lambda: tgp.posterior_marginals(dm, ym, Rn))
    assert any(n.startswith("k_compose_smoother") for n in names) == (d >= 6), names


@pytest.mark.parametrize("d,p", [(6, 2), (7, 3)])
def test_split_smoother_vector_observations(tgp, d, p):
    """p > 1: the filtered-state scratch holds one state per scalar micro-step; the compose kernel reads the state before each
    time step's predict (the last micro-step of the step before)"""
    rng = np.random.default_rng(13 * d + p)
    T = 650
    model = U.random_lgssm_small(rng, False, d, p, T)
    eps = (rng.standard_normal((T, d)), rng.standard_normal((T, p)), rng.standard_normal(d))
    y = ref.rand(model, *eps)
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    dm = tgp.LGSSM(tr, tgp.SmallOutputLGC(model["H"], model["h"], np.diagonal(model["R"], axis1=-2, axis2=-1)), T=T)
    hd = dm.handle()
    hd.set_option(tgp._lib.OPT_GROUP, 0)
    post = ref.posterior(model, y)
    Rn = rng.random((T, p)) * 0.1
    pm, pC = ref.marginals(ref.replace_observation_noise_cov(post, np.stack([np.diag(v) for v in Rn])))
    for chunk in (0, 4, 11):
        hd.set_option(tgp._lib.OPT_CHUNK, chunk)
        hd.set_option(tgp._lib.OPT_SPLIT_SMOOTHER, 0)
        fm, fv = tgp.posterior_marginals(dm, y, Rn)
        hd.set_option(tgp._lib.OPT_SPLIT_SMOOTHER, 1)
        (sm, sv), names = _names(hd, tgp, lambda: tgp.posterior_marginals(dm, y, Rn))
        assert any(n.startswith("k_compose_smoother") for n in names), names
        np.testing.assert_allclose(sm, fm, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(sv, fv, rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(sm, pm, rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(sv, np.diagonal(pC, axis1=-2, axis2=-1), rtol=1e-8, atol=1e-9)


def test_split_smoother_option_range(tgp):
    rng = np.random.default_rng(3)
    model = U.random_lgssm(rng, False, 6, 50)
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], model["h"], model["R"]), T=50)
    with pytest.raises(Exception):
        dm.handle().set_option(tgp._lib.OPT_SPLIT_SMOOTHER, 3)
