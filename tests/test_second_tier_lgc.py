"""LargeOutputLGC / BottleneckLGC.
CPU tier: the oracle's literal restatements (oracle/lgssm_ref.py, lgc.jl:179-204, 305-336) against the reference's own
consistency tests (test/models/linear_gaussian_conditionals.jl:65-75 "LargeOutputLGC consistency with SmallOutputLGC",
:156-167 "BottleneckLGC consistency with SmallOutputLGC", missing data :77-91, :169-185) at the reference's tolerances.
GPU tier: LGSSMs with these emissions through the C ABI (the device treats both as p scalar updates per time step)
against the oracle's literal Large / Bottleneck recursions, again at the reference's consistency tolerances."""
import numpy as np
import pytest

from oracle import lgssm_ref as ref
from tests import _util as U


def _psd(rng, k, lo=0.5, hi=1.5):
    Q = np.linalg.qr(rng.standard_normal((k, k)))[0]
    return (Q * (rng.random(k) * (hi - lo) + lo)) @ Q.T


@pytest.mark.parametrize("Dlat,Dobs", [(1, 1), (3, 1), (1, 2), (3, 2), (2, 5), (3, 7)])
@pytest.mark.parametrize("diag", [True, False])
def test_large_equals_small(Dlat, Dobs, diag):
    rng = np.random.default_rng(10 * Dlat + Dobs + diag)
    m, P = rng.standard_normal(Dlat), _psd(rng, Dlat)
    A, a = rng.standard_normal((Dobs, Dlat)), rng.standard_normal(Dobs)
    Q = np.diag(rng.random(Dobs) + 0.1) if diag else _psd(rng, Dobs)
    y = A @ m + a + rng.standard_normal(Dobs)
    ms, Ps, ls = ref.posterior_and_lml_small(m, P, A, a, Q, y)
    ml, Pl, ll = ref.posterior_and_lml_large(m, P, A, a, Q, y)
    np.testing.assert_allclose(ml, ms, rtol=1.5e-8, atol=1e-9)      # Julia isapprox default rtol = sqrt(eps)
    np.testing.assert_allclose(Pl, Ps, rtol=1.5e-8, atol=1e-9)
    assert abs(ll - ls) <= 1.5e-8 * abs(ls) + 1e-9


@pytest.mark.parametrize("Din,Dmid,Dout", [(1, 1, 1), (3, 1, 2), (3, 3, 2), (2, 1, 5), (4, 3, 6)])
@pytest.mark.parametrize("diag", [True, False])
def test_bottleneck_equals_composed_small(Din, Dmid, Dout, diag):
    rng = np.random.default_rng(100 * Din + 10 * Dmid + Dout + diag)
    m, P = rng.standard_normal(Din), _psd(rng, Din)
    Hb, hb = rng.standard_normal((Dmid, Din)), rng.standard_normal(Dmid)
    A, a = rng.standard_normal((Dout, Dmid)), rng.standard_normal(Dout)
    Q = np.diag(rng.random(Dout) + 0.1) if diag else _psd(rng, Dout)
    y = A @ (Hb @ m + hb) + a + rng.standard_normal(Dout)
    ms, Ps, ls = ref.posterior_and_lml_small(m, P, A @ Hb, A @ hb + a, Q, y)
    mb, Pb, lb = ref.posterior_and_lml_bottleneck(m, P, Hb, hb, A, a, Q, y)
    np.testing.assert_allclose(mb, ms, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(Pb, Ps, rtol=1e-6, atol=1e-8)
    assert abs(lb - ls) <= 1e-6 * abs(ls) + 1e-8


def _bottleneck_model(rng, tv, d, dz, p, T):
    model = U.random_lgssm_small(rng, tv, dz, p, T)            # fan-out (H, h, R) acts on the dz-dimensional projection
    base = U.random_lgssm(rng, tv, d, T)
    n = T if tv else 1
    for k in ("A", "a", "Q", "x0m", "x0P"):
        model[k] = base[k]
    model.update(kind="bottleneck", Hb=rng.standard_normal((n, dz, d)), hb=rng.standard_normal((n, dz)))
    return model


@pytest.mark.parametrize("tv", [True, False])
def test_lgssm_large_and_bottleneck_oracle_consistency(tv):
    """whole-model version of the two consistency tests (logpdf, filter) incl. missing data with diagonal noise"""
    rng = np.random.default_rng(5 + tv)
    T = 60
    small = U.random_lgssm_small(rng, tv, 2, 5, T)
    y = ref.rand(small, rng.standard_normal((T, 2)), rng.standard_normal((T, 5)), rng.standard_normal(2))
    large = dict(small, kind="large")
    assert abs(ref.logpdf(large, y) - ref.logpdf(small, y)) <= 1e-7 * abs(ref.logpdf(small, y))
    miss = rng.random((T, 5)) < 0.2
    assert abs(ref.logpdf_missing(large, y, miss) - ref.logpdf_missing(small, y, miss)) <= 1e-7 * abs(ref.logpdf_missing(small, y, miss))
    bott = _bottleneck_model(rng, tv, 4, 2, 5, T)
    yb = ref.rand(bott, rng.standard_normal((T, 4)), rng.standard_normal((T, 5)), rng.standard_normal(4))
    comp = ref.small_from_bottleneck(bott)
    np.testing.assert_allclose(ref.rand(comp, np.zeros((T, 4)), np.zeros((T, 5)), np.zeros(4)),
                               ref.rand(bott, np.zeros((T, 4)), np.zeros((T, 5)), np.zeros(4)), rtol=1e-12, atol=1e-12)
    assert abs(ref.logpdf(bott, yb) - ref.logpdf(comp, yb)) <= 1e-6 * abs(ref.logpdf(comp, yb))
    fm_b, fP_b = ref.filter_(bott, yb)
    fm_c, fP_c = ref.filter_(comp, yb)
    np.testing.assert_allclose(fm_b, fm_c, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(fP_b, fP_c, rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------------ GPU tier
@pytest.fixture(scope="module")
def tgp():
    import temporalgps_jl_amd as t
    t._lib.load()
    return t


def _transitions(tgp, model):
    return tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))


@pytest.mark.gpu
@pytest.mark.parametrize("d,p", [(2, 5), (3, 7), (1, 4)])
@pytest.mark.parametrize("tv", [True, False])
def test_gpu_large_output_lgc(tgp, d, p, tv):
    rng = np.random.default_rng(31 * d + p + tv)
    T = 500
    model = dict(U.random_lgssm_small(rng, tv, d, p, T), kind="large")
    y = ref.rand(dict(model, kind="small"), rng.standard_normal((T, d)), rng.standard_normal((T, p)), rng.standard_normal(d))
    dm = tgp.LGSSM(_transitions(tgp, model), tgp.LargeOutputLGC(model["H"], model["h"], np.diagonal(model["R"], axis1=-2, axis2=-1)), T=T)
    dm.handle().set_option(tgp._lib.OPT_CHUNK, 3)
    lp = ref.logpdf(model, y)                                   # the oracle's literal LargeOutputLGC recursion
    assert abs(tgp.logpdf(dm, y) - lp) <= 1e-7 * abs(lp)
    fm, fP = ref.filter_(model, y)
    m, P = tgp._filter(dm, y)
    np.testing.assert_allclose(m, fm, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(P, fP, rtol=1e-6, atol=1e-7)
    miss = rng.random((T, p)) < 0.2                             # per-element missing, diagonal noise (lgc.jl:209-217)
    ym = y.copy()
    ym[miss] = np.nan
    lpm = ref.logpdf_missing(model, y, miss)
    assert abs(tgp.logpdf(dm, ym) - lpm) <= 1e-7 * abs(lpm)
    Rn = rng.random((T, p)) * 0.1
    post = ref.posterior(model, y)
    pm, pC = ref.marginals(ref.replace_observation_noise_cov(post, np.stack([np.diag(v) for v in Rn])))
    gm, gv = tgp.posterior_marginals(dm, y, Rn)
    np.testing.assert_allclose(gm, pm, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(gv, np.diagonal(pC, axis1=-2, axis2=-1), rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("d,dz,p", [(4, 2, 5), (3, 1, 4), (6, 3, 6)])
@pytest.mark.parametrize("tv", [True, False])
def test_gpu_bottleneck_lgc(tgp, d, dz, p, tv):
    rng = np.random.default_rng(7 * d + 3 * dz + p + tv)
    T = 400
    model = _bottleneck_model(rng, tv, d, dz, p, T)
    y = ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal((T, p)), rng.standard_normal(d))
    fan_out = tgp.LargeOutputLGC(model["H"], model["h"], np.diagonal(model["R"], axis1=-2, axis2=-1))
    dm = tgp.LGSSM(_transitions(tgp, model), tgp.BottleneckLGC(model["Hb"], model["hb"], fan_out), T=T)
    dm.handle().set_option(tgp._lib.OPT_CHUNK, 3)
    lp = ref.logpdf(model, y)                                   # the oracle's literal BottleneckLGC recursion
    assert abs(tgp.logpdf(dm, y) - lp) <= 1e-6 * abs(lp)
    fm, fP = ref.filter_(model, y)
    m, P = tgp._filter(dm, y)
    np.testing.assert_allclose(m, fm, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(P, fP, rtol=1e-5, atol=1e-6)
    mm, mC = ref.marginals(model)                              # prior marginals of the observations (lgc.jl:314-318)
    gm, gv = tgp.marginals(dm)
    np.testing.assert_allclose(gm, mm, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(gv, np.diagonal(mC, axis1=-2, axis2=-1), rtol=1e-8, atol=1e-8)
    eps = (rng.standard_normal((T, d)), rng.standard_normal((T, p)), rng.standard_normal(d))
    np.testing.assert_allclose(tgp.rand(eps, dm), ref.rand(model, *eps), rtol=1e-8, atol=1e-8)
