"""The C restatement (oracle/seq_kalman.c) must agree with the NumPy restatement (oracle/lgssm_ref.py)."""
import numpy as np
import pytest

from oracle import components as oc
from oracle import lgssm_ref as ref
from oracle import seq_kalman as sk

CASES = [
    (("matern12",), ("regular", 0.0, 0.1, 200), 0.1),
    (("matern32",), ("regular", 0.0, 0.1, 200), 0.1),
    (("matern52",), ("regular", 0.0, 0.1, 200), 0.1),
    (("sum", ("matern52",), ("matern32",)), ("regular", 0.0, 0.1, 150), 0.1),
    (("sum", ("matern52",), ("matern52",)), ("regular", 0.0, 0.1, 100), 0.2),
    (("scaled", 1.5, ("stretched", 0.7, ("matern52",))), None, None),   # irregular + hetero
    (("sum", ("matern52",), ("matern12",)), None, None),
]


def _case(i):
    rng = np.random.default_rng(100 + i)
    k, t, s2 = CASES[i]
    if t is None:
        t = np.cumsum(rng.random(120) * 0.1 + 0.05)
        s2 = rng.random(120) * 0.2 + 0.05
    model = oc.build_lgssm(k, t, s2)
    T, d = model["T"], len(model["x0m"])
    eps = rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d)
    y = ref.rand(model, *eps)
    return model, y, eps


@pytest.mark.parametrize("i", range(len(CASES)))
def test_c_matches_numpy(i):
    model, y, eps = _case(i)
    np.testing.assert_allclose(sk.rand(model, *eps), y, rtol=1e-12, atol=1e-12)
    lp = ref.logpdf(model, y)
    lml, ms, Ps = sk.filter_(model, y, want_states=True)
    assert abs(lml - lp) <= 1e-12 * abs(lp)
    rm, rP = ref.filter_(model, y)
    np.testing.assert_allclose(ms, rm, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(Ps, rP, rtol=1e-11, atol=1e-13)
    post_c, post = sk.posterior(model, y), ref.posterior(model, y)
    for key in ("A", "a", "Q", "x0m", "x0P"):
        np.testing.assert_allclose(post_c[key], post[key], rtol=1e-9, atol=1e-11)
    Rn = np.full(model["T"], 1e-18)
    mean, var = sk.posterior_marginals(model, y, Rn)
    rmean, rvar = ref.marginals(ref.replace_observation_noise_cov(post, Rn))
    np.testing.assert_allclose(mean, rmean, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(var, rvar, rtol=1e-9, atol=1e-12)
    pm, pv = sk.prior_marginals(model)
    qm, qv = ref.marginals(model)
    np.testing.assert_allclose(pm, qm, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(pv, qv, rtol=1e-12)


def test_openmp_host_build_of_the_chunk_functions_compiles():
    """bench.py's all-core CPU leg (oracle/omp_scan.py) compiles tests/hostsim/hostsim.cpp with -fopenmp; a pragma in front of
    a non-loop statement only shows up in that build (the GPU box rebuilds it: file times differ there)."""
    import shutil
    import pytest
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    from oracle import omp_scan
    so = omp_scan.build(force=True)
    import os
    assert os.path.exists(so)
