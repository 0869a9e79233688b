"""CPU tier: the stationary-covariance steps of passes 2 / 3 (tgp_chunk_body.inc, "stationary covariance") through the host build of
the engine's own headers (tests/hostsim). A shared-layout model with one noise variance, scalar observations and no missing data
lets a chunk switch to mean-only steps once its covariance repeats with period 2 bit for bit; the claim under test is that the
switch changes NO bit of any result -- log-likelihood, filtering distributions, posterior marginals -- and that it is actually
taken. The models are the reference's own kernels (d = 1, 2, 3) on a regular grid."""
import os

import numpy as np
import pytest

from oracle import components as oc
from oracle import seq_kalman as sk
from tests import _util as U

SPECS = {1: ("matern12",), 2: ("matern32",), 3: ("matern52",)}


def _case(d, T, dt, seed):
    model = oc.build_lgssm(SPECS[d], ("regular", 0.0, dt, T), 0.1)
    rng = np.random.default_rng(seed)
    y = sk.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    return model, y


def _run(model, what, y, L0, steady, **kw):
    old = os.environ.pop("HOSTSIM_STEADY", None)
    try:
        if steady:
            os.environ["HOSTSIM_STEADY"] = "2"       # "2": also report how many steps ran mean-only (stderr)
        return U.hostsim_run(model, what, y=y, L0=L0, BS=3, **kw)
    finally:
        os.environ.pop("HOSTSIM_STEADY", None)
        if old is not None:
            os.environ["HOSTSIM_STEADY"] = old


def _mean_only_steps(err):
    lines = [ln for ln in err.splitlines() if ln.startswith("hostsim steady:")]
    assert lines, err
    return int(lines[-1].split()[2])


@pytest.mark.parametrize("d", [1, 2, 3])
@pytest.mark.parametrize("L0", [37, 64, 153])
def test_posterior_marginals_bit_identical_and_taken(d, L0, capfd):
    T = 2500
    model, y = _case(d, T, 0.1, seed=10 * d + L0)
    Rn = np.array([1e-3])
    a = _run(model, 2, y, L0, False, Rnew=Rn)
    capfd.readouterr()
    b = _run(model, 2, y, L0, True, Rnew=Rn)
    n_fast = _mean_only_steps(capfd.readouterr().err)
    assert a["rc"] == 0 and b["rc"] == 0
    assert a["lml"] == b["lml"]
    assert np.array_equal(a["mean"], b["mean"]) and np.array_equal(a["var"], b["var"])
    assert np.array_equal(a["xfm"], b["xfm"]) and np.array_equal(a["xfP"], b["xfP"])
    # most of the series runs mean-only: every chunk but the first settles within ~15 steps
    assert n_fast > 0.4 * T, n_fast
    # ... and both agree with the sequential restatement of the reference
    lp = sk.logpdf(model, y)
    assert abs(a["lml"] - lp) <= 1e-10 * abs(lp)
    mean, var = sk.posterior_marginals(model, y, Rn)
    np.testing.assert_allclose(b["mean"], mean, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(b["var"], var, rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("d", [1, 2, 3])
def test_logpdf_and_filter_bit_identical(d):
    model, y = _case(d, 1800, 0.05, seed=3 + d)
    for what in (0, 1):
        a = _run(model, what, y, 50, False)
        b = _run(model, what, y, 50, True)
        assert a["lml"] == b["lml"]
        if what == 1:
            assert np.array_equal(a["m"], b["m"]) and np.array_equal(a["P"], b["P"])


def test_ragged_last_chunk_and_short_series():
    # the last chunk is shorter than the others (and may end before or after its covariance settles); a series shorter than one chunk
    for T, L0 in ((1000, 153), (1000, 96), (1013, 40), (30, 64), (157, 153)):
        model, y = _case(3, T, 0.1, seed=T)
        Rn = np.array([0.2])
        a = _run(model, 2, y, L0, False, Rnew=Rn)
        b = _run(model, 2, y, L0, True, Rnew=Rn)
        assert a["lml"] == b["lml"], (T, L0)
        assert np.array_equal(a["mean"], b["mean"]) and np.array_equal(a["var"], b["var"]), (T, L0)


def test_per_step_new_noise_and_reverse_ordering():
    # per-step R_new only enters the emission of pass 3 (var = v + R_new[t]); Reverse-ordered priors skip the first predict
    model, y = _case(2, 900, 0.1, seed=5)
    Rn = np.random.default_rng(1).random(900) + 0.1
    a = _run(model, 2, y, 64, False, Rnew=Rn)
    b = _run(model, 2, y, 64, True, Rnew=Rn)
    assert np.array_equal(a["mean"], b["mean"]) and np.array_equal(a["var"], b["var"])
    rmodel = dict(model, ordering="R")
    a = _run(rmodel, 0, y, 64, False)
    b = _run(rmodel, 0, y, 64, True)
    assert a["lml"] == b["lml"]


def test_not_taken_where_the_step_is_not_the_same_map(capfd):
    # missing data changes the noise variance of single steps: the option must leave such a series alone
    model, y = _case(3, 800, 0.1, seed=9)
    miss = np.zeros(800, dtype=np.uint8)
    miss[[5, 300, 301, 650]] = 1
    a = _run(model, 2, y, 64, False, missing=miss, Rnew=np.array([0.1]))
    capfd.readouterr()
    b = _run(model, 2, y, 64, True, missing=miss, Rnew=np.array([0.1]))
    assert "hostsim steady:" not in capfd.readouterr().err
    assert a["lml"] == b["lml"] and np.array_equal(a["mean"], b["mean"]) and np.array_equal(a["var"], b["var"])
