"""GPU tier: vector observations (SmallOutputLGC, p > 1) through the C ABI against the oracle's joint update
(oracle/lgssm_ref.py, lgc.jl:129-141). Tolerances as in test_gpu_parity.py."""
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


def to_device(tgp, model, diag=True):
    tr = tgp.GaussMarkovModel(tgp.Forward if model["ordering"] == "F" else tgp.Reverse, model["A"], model["a"], model["Q"],
                              tgp.Gaussian(model["x0m"], model["x0P"]))
    R = np.diagonal(model["R"], axis1=-2, axis2=-1) if diag else model["R"]
    return tgp.LGSSM(tr, tgp.SmallOutputLGC(model["H"], model["h"], R), T=model["T"])


@pytest.mark.parametrize("d,p", [(2, 2), (3, 2), (4, 3), (3, 5), (6, 2)])
@pytest.mark.parametrize("tv", [True, False])
@pytest.mark.parametrize("ordering", ["F", "R"])
def test_vector_obs_diag_noise(tgp, d, p, tv, ordering):
    rng = np.random.default_rng(100 * d + 10 * p + tv)
    T = 700
    model = U.random_lgssm_small(rng, tv, d, p, T, ordering)
    eps = (rng.standard_normal((T, d)), rng.standard_normal((T, p)), rng.standard_normal(d))
    y = ref.rand(model, *eps)
    dm = to_device(tgp, model)
    dm.handle().set_option(tgp._lib.OPT_CHUNK, 3)            # rounded up to whole time steps; several scan levels
    lp = ref.logpdf(model, y)
    assert abs(tgp.logpdf(dm, y) - lp) <= 1e-10 * abs(lp)
    fm, fP = ref.filter_(model, y)
    m, P = tgp._filter(dm, y)
    np.testing.assert_allclose(m, fm, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(P, fP, rtol=1e-8, atol=1e-9)
    mm, mC = ref.marginals(model)
    gm, gv = tgp.marginals(dm)
    np.testing.assert_allclose(gm, mm, rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(gv, np.diagonal(mC, axis1=-2, axis2=-1), rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(tgp.rand(eps, dm), y, rtol=1e-9, atol=1e-9)
    if ordering == "F":
        post = ref.posterior(model, y)
        dpost = tgp.posterior(dm, y)
        np.testing.assert_allclose(dpost.transitions.As, post["A"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(dpost.transitions.Qs, post["Q"], rtol=1e-8, atol=1e-9)
        Rn = rng.random((T, p)) * 0.1
        pm, pC = ref.marginals(ref.replace_observation_noise_cov(post, np.stack([np.diag(v) for v in Rn])))
        gm, gv = tgp.posterior_marginals(dm, y, Rn)
        np.testing.assert_allclose(gm, pm, rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(gv, np.diagonal(pC, axis1=-2, axis2=-1), rtol=1e-8, atol=1e-9)
        gm2, gv2 = tgp.marginals(tgp.replace_observation_noise_cov(dpost, Rn))     # materialised Reverse model, p > 1
        np.testing.assert_allclose(gm2, pm, rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(gv2, np.diagonal(pC, axis1=-2, axis2=-1), rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("tv", [True, False])
def test_vector_obs_dense_noise_whitened(tgp, tv):
    rng = np.random.default_rng(77 + tv)
    T, d, p = 400, 3, 3
    model = U.random_lgssm_small(rng, tv, d, p, T, dense_R=True)
    y = rng.standard_normal((T, p))
    dm = to_device(tgp, model, diag=False)
    lp = ref.logpdf(model, y)
    assert abs(tgp.logpdf(dm, y) - lp) <= 1e-10 * abs(lp)
    fm, fP = ref.filter_(model, y)
    m, P = tgp._filter(dm, y)
    np.testing.assert_allclose(m, fm, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(P, fP, rtol=1e-8, atol=1e-9)
    missing = rng.random(T) < 0.2
    ym = y.copy()
    ym[missing] = np.nan
    lpm = ref.logpdf_missing(model, y, missing)
    assert abs(tgp.logpdf(dm, ym) - lpm) <= 1e-10 * abs(lpm)
    # prior marginals and samples do not involve the update: the correlated noise enters additively (lgc.jl:46-52, :84-87)
    mm, mC = ref.marginals(model)
    gm, gv = tgp.marginals(dm)
    np.testing.assert_allclose(gm, mm, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(gv, np.diagonal(mC, axis1=-2, axis2=-1), rtol=1e-9, atol=1e-9)       # marginals_diag (lgssm.jl:128-137)
    eps = (rng.standard_normal((T, d)), rng.standard_normal((T, p)), rng.standard_normal(d))
    np.testing.assert_allclose(tgp.rand(eps, dm), ref.rand(model, *eps), rtol=1e-9, atol=1e-9)
    import torch
    eps_dev = (torch.as_tensor(eps[0], device="cuda:0"), torch.as_tensor(eps[1], device="cuda:0"), eps[2])
    np.testing.assert_allclose(tgp.rand(eps_dev, dm).cpu().numpy(), ref.rand(model, *eps), rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("per_element", [False, True])
def test_vector_obs_missing(tgp, per_element):
    rng = np.random.default_rng(9)
    T, d, p = 500, 3, 3
    model = U.random_lgssm_small(rng, True, d, p, T)
    y = rng.standard_normal((T, p))
    missing = rng.random((T, p)) < 0.3 if per_element else rng.random(T) < 0.3
    ym = y.copy()
    ym[missing] = np.nan
    dm = to_device(tgp, model)
    lp = ref.logpdf_missing(model, y, missing)
    assert abs(tgp.logpdf(dm, ym) - lp) <= 1e-10 * abs(lp)
    fm, fP = ref.filter_missing(model, y, missing)
    m, P = tgp._filter(dm, ym)
    np.testing.assert_allclose(m, fm, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(P, fP, rtol=1e-8, atol=1e-9)


def test_small_output_p1_dense_noise_is_whitened(tgp):
    """A SmallOutputLGC with ONE output and a 1 x 1 dense noise matrix: handle() whitens H, h and R, so y must be whitened as
    well (it was not: logpdf / _filter were silently wrong for this input)."""
    rng = np.random.default_rng(5)
    T, d, p = 300, 3, 1
    model = U.random_lgssm_small(rng, False, d, p, T, dense_R=True)
    y = rng.standard_normal((T, p))
    dm = to_device(tgp, model, diag=False)
    lp = ref.logpdf(model, y)
    assert abs(tgp.logpdf(dm, y) - lp) <= 1e-10 * abs(lp)
    assert abs(tgp.logpdf(dm, y[:, 0]) - lp) <= 1e-10 * abs(lp)
    fm, fP = ref.filter_(model, y)
    m, P = tgp._filter(dm, y)
    np.testing.assert_allclose(m, fm, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(P, fP, rtol=1e-8, atol=1e-9)
    mm, mC = ref.marginals(model)
    gm, gv = tgp.marginals(dm)
    np.testing.assert_allclose(np.asarray(gm).reshape(T), mm.reshape(T), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(np.asarray(gv).reshape(T), mC.reshape(T), rtol=1e-9, atol=1e-9)
    eps = (rng.standard_normal((T, d)), rng.standard_normal((T, p)), rng.standard_normal(d))
    np.testing.assert_allclose(np.asarray(tgp.rand(eps, dm)).reshape(T), np.asarray(ref.rand(model, *eps)).reshape(T), rtol=1e-9, atol=1e-9)


def test_dense_noise_with_observations_on_the_device(tgp):
    """The same whitening when y is a CUDA tensor (it was skipped there too)."""
    import torch
    rng = np.random.default_rng(6)
    T, d, p = 200, 3, 3
    model = U.random_lgssm_small(rng, True, d, p, T, dense_R=True)
    y = rng.standard_normal((T, p))
    dm = to_device(tgp, model, diag=False)
    lp = ref.logpdf(model, y)
    yd = torch.as_tensor(y, device="cuda:0")
    assert abs(tgp.logpdf(dm, yd) - lp) <= 1e-10 * abs(lp)
    missing = rng.random(T) < 0.2
    lpm = ref.logpdf_missing(model, y, missing)
    md = torch.as_tensor(np.repeat(missing[:, None], p, axis=1), device="cuda:0")
    assert abs(tgp.logpdf(dm, (yd, md)) - lpm) <= 1e-10 * abs(lpm)


@pytest.mark.parametrize("d,p", [(5, 2), (6, 3), (8, 2), (9, 4), (16, 3), (16, 5)])
@pytest.mark.parametrize("ordering", ["F", "R"])
@pytest.mark.parametrize("per", ["A", "ah", "AaQ"])
def test_vector_obs_partly_shared_blocks_share_the_noise_diagonal(tgp, d, p, ordering, per):
    """Per-step transitions (or offsets) with ONE shared emission block whose noise diagonal has p different entries: the group-per-chunk
    passes (d >= 5 in the per-step layout) took R[0] for every row of such a model until round 3 (scripts/stress_general2.py) -- the
    all-per-step and all-shared models of the tests above never reach that branch."""
    rng = np.random.default_rng(1000 * d + 10 * p + (ordering == "R") + len(per))
    T = 300
    tv = U.random_lgssm_small(rng, True, d, p, T, ordering)
    model = dict(tv)
    for k_ in ("A", "a", "Q", "H", "h", "R"):
        if k_ not in per:
            model[k_] = tv[k_][:1]
    model["R"] = np.diag(rng.random(p) * 2.0 + 0.05)[None]              # clearly different entries
    eps = (rng.standard_normal((T, d)), rng.standard_normal((T, p)), rng.standard_normal(d))
    y = ref.rand(model, *eps)
    dm = to_device(tgp, model)
    lp = ref.logpdf(model, y)
    assert abs(tgp.logpdf(dm, y) - lp) <= 1e-10 * abs(lp)
    fm, fP = ref.filter_(model, y)
    m, P = tgp._filter(dm, y)
    np.testing.assert_allclose(m, fm, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(P, fP, rtol=1e-8, atol=1e-9)
    mm, mC = ref.marginals(model)
    gm, gv = tgp.marginals(dm)
    np.testing.assert_allclose(gm, mm, rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(gv, np.diagonal(mC, axis1=-2, axis2=-1), rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(tgp.rand(eps, dm), y, rtol=1e-8, atol=1e-8)
    if ordering == "F":
        post = ref.posterior(model, y)
        Rn = rng.random((T, p)) * 0.1
        pm, pC = ref.marginals(ref.replace_observation_noise_cov(post, np.stack([np.diag(v) for v in Rn])))
        gm, gv = tgp.posterior_marginals(dm, y, Rn)
        np.testing.assert_allclose(gm, pm, rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(gv, np.diagonal(pC, axis1=-2, axis2=-1), rtol=1e-8, atol=1e-9)
