"""Separable space-time GPs (reference: test/space_time/to_gauss_markov.jl:30-90).
CPU: the oracle's literal restatement == dense GP with the separable kernel (pins the oracle), the product's
host construction == the oracle's. GPU: the literal device model and the eigen-decoupled shortcut == oracle."""
import numpy as np
import pytest

from oracle import components as oc
from oracle import dense_gp as dg
from oracle import lgssm_ref as ref

CASES = [
    (("se",), ("matern32",), 3, ("regular", 0.0, 0.3, 11), "homo"),
    (("se",), ("matern52",), 2, ("regular", 0.0, 0.3, 9), "hetero"),
    (("scaled", 1.3, ("matern32",)), ("stretched", 0.7, ("matern32",)), 4, None, "homo"),
]


def _case(i):
    rng = np.random.default_rng(50 + i)
    ks, kt, Nr, t, noise = CASES[i]
    r = np.sort(rng.standard_normal(Nr))
    if t is None:
        t = np.cumsum(rng.random(10) * 0.3 + 0.1)
    T = oc.n_times(t)
    s2 = 0.1 if noise == "homo" else rng.random((T, Nr)) * 0.2 + 0.05
    y = rng.standard_normal(T * Nr)
    return ks, kt, r, t, T, Nr, s2, y


@pytest.mark.parametrize("i", range(len(CASES)))
def test_oracle_space_time_equals_dense_gp(i):
    ks, kt, r, t, T, Nr, s2, y = _case(i)
    model = oc.build_lgssm_separable(ks, kt, r, t, s2)
    K = dg.separable_kernelmatrix(ks, kt, r, oc.times(t))
    noise = np.full(T * Nr, s2) if np.ndim(s2) == 0 else np.asarray(s2).reshape(-1)
    lp = ref.logpdf(model, y.reshape(T, Nr))
    lp_d = dg.mvn_logpdf(K + np.diag(noise), y)
    assert abs(lp - lp_d) <= 1.5e-8 * abs(lp_d)
    mm, mC = ref.marginals(model)
    np.testing.assert_allclose(np.diagonal(mC, axis1=-2, axis2=-1).reshape(-1), np.diag(K) + noise, rtol=1.5e-8)
    post = ref.replace_observation_noise_cov(ref.posterior(model, y.reshape(T, Nr)), np.stack([0.1 * np.eye(Nr)] * T))
    pm, pC = ref.marginals(post)
    mu_d, var_d = dg.mvn_posterior_marginals(K, noise, y, 0.1)
    np.testing.assert_allclose(pm.reshape(-1), mu_d, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(np.diagonal(pC, axis1=-2, axis2=-1).reshape(-1), var_d, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("i", range(len(CASES)))
def test_product_host_construction_matches_oracle(i):
    import temporalgps_jl_amd  # noqa: F401
    from temporalgps_jl_amd import lti_sde, space_time
    ks, kt, r, t, T, Nr, s2, y = _case(i)
    tk = lambda spec: space_time.SEKernel() if spec == ("se",) else lti_sde.to_kernel(spec)
    tt = lti_sde.RegularSpacing(t[1], t[2], t[3]) if oc.is_regular(t) else t
    A, a, Q, H, h, (m0, P0) = space_time.lgssm_components(space_time.Separable(tk(ks), tk(kt)), space_time.RectilinearGrid(r, tt))
    m = oc.build_lgssm_separable(ks, kt, r, t, s2)
    for got, want in ((A, m["A"]), (Q, m["Q"]), (H, m["H"]), (h, m["h"]), (m0, m["x0m"]), (P0, m["x0P"])):
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(CASES)))
def test_hip_space_time_literal_and_decoupled(i):
    import temporalgps_jl_amd as tgp
    from temporalgps_jl_amd import lti_sde, space_time
    ks, kt, r, t, T, Nr, s2, y = _case(i)
    tk = lambda spec: space_time.SEKernel() if spec == ("se",) else lti_sde.to_kernel(spec)
    tt = lti_sde.RegularSpacing(t[1], t[2], t[3]) if oc.is_regular(t) else t
    k, grid = space_time.Separable(tk(ks), tk(kt)), space_time.RectilinearGrid(r, tt)
    model = oc.build_lgssm_separable(ks, kt, r, t, s2)
    Y = y.reshape(T, Nr)
    lp = ref.logpdf(model, Y)
    post = ref.replace_observation_noise_cov(ref.posterior(model, Y), np.stack([0.1 * np.eye(Nr)] * T))
    pm, pC = ref.marginals(post)
    pv = np.diagonal(pC, axis1=-2, axis2=-1)
    d = len(model["x0m"])
    if d <= 8:      # the reference's own (dense, Nr * d_t-dimensional) model on the device
        dm = space_time.build_lgssm(k, grid, s2)
        assert abs(tgp.logpdf(dm, Y) - lp) <= 1e-10 * abs(lp)
        gm, gv = tgp.posterior_marginals(dm, Y, np.full((1, Nr), 0.1))
        np.testing.assert_allclose(gm, pm, rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(gv, pv, rtol=1e-8, atol=1e-9)
    if np.ndim(s2) == 0:   # eigen-decoupled shortcut: noise equal across space
        dec = space_time.DecoupledSpaceTime(k, grid, s2)
        assert abs(dec.logpdf(y) - lp) <= 1e-9 * abs(lp)
        gm, gv = dec.posterior_marginals(y, 0.1)
        np.testing.assert_allclose(gm, pm, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(gv, pv, rtol=1e-7, atol=1e-9)
    else:
        with pytest.raises(ValueError):
            space_time.DecoupledSpaceTime(k, grid, s2)
