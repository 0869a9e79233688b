"""CPU tier: the dense-powers one-launch smoother (logpdf + posterior marginals of LTI models WITHOUT a modal form -- a sum of two
kernels with one length scale has a defective closed loop) run on the host -- the product's own plan and head functions
(tgp_steady_plan.hpp build_smooth / smooth_head_*) and a lane-by-lane restatement of k_smooth_one's orchestration
(tests/hostsim/smoothsim.cpp) -- against the oracle's literal restatement of the reference's sequential recursions (lgssm.jl:99-238).
Tolerances (fp64): logpdf rel 1e-10; posterior marginals abs 1e-8 * scale."""
import numpy as np
import pytest

from oracle import lgssm_ref as ref
from tests import _util as U

CASES = [
    (("sum", ("matern52",), ("matern52",)), 0.1, 0.1),          # SURVEY 8d's cfg3 at d = 6: no modal form
    (("sum", ("matern32",), ("matern32",)), 0.1, 0.2),
    (("sum", ("matern52",), ("matern52",), ("matern32",)), 0.1, 0.1),      # d = 8
    (("matern52",), 0.1, 0.1),                                  # (models WITH a modal form run through it just as well)
    (("sum", ("matern52",), ("matern32",)), 0.05, 0.3),
    (("matern12",), 0.2, 0.5),
    (("sum", ("matern12",), ("matern12",)), 0.1, 0.1),
]


def _reference(model, y, Rn):
    lp = ref.logpdf(model, y)
    post = ref.posterior(model, y)
    pm, pv = ref.marginals(ref.replace_observation_noise_cov(post, np.broadcast_to(Rn, (model["T"],)).copy()))
    return lp, pm, pv


def _check(r, lp, pm, pv):
    assert r["rc"] == 0 and r["why"] == 0, r
    assert abs(r["lml"] - lp) <= 1e-10 * abs(lp), (r["lml"], lp)
    sc = max(1.0, np.abs(pm).max())
    assert np.abs(r["mean"] - pm).max() <= 1e-8 * sc, np.abs(r["mean"] - pm).max()
    assert np.abs(r["var"] - pv).max() <= 1e-8 * max(1.0, pv.max()), np.abs(r["var"] - pv).max()


@pytest.mark.parametrize("i", range(len(CASES)))
@pytest.mark.parametrize("T", [4200, 9000])
def test_dense_powers_smoother_equals_oracle(i, T):
    k, dt, s2 = CASES[i]
    model, y, _ = U.gp_case(k, ("regular", 0.0, dt, T), s2, seed=i)
    Rn = 0.05
    lp, pm, pv = _reference(model, y, Rn)
    r = U.smoothsim_run(model, y, Rn)
    _check(r, lp, pm, pv)
    assert r["nwg"] >= 2      # (several spans: the halos and the chain over the tiles are exercised)


def test_per_step_new_noise_and_a_series_ending_inside_a_tile():
    k, dt, s2 = CASES[0]
    T = 5003
    model, y, _ = U.gp_case(k, ("regular", 0.0, dt, T), s2, seed=11)
    Rn = np.random.default_rng(3).random(T) * 0.1
    lp, pm, pv = _reference(model, y, Rn)
    _check(U.smoothsim_run(model, y, Rn), lp, pm, pv)


def test_a_series_too_short_is_declined():
    k, dt, s2 = CASES[0]
    model, y, _ = U.gp_case(k, ("regular", 0.0, dt, 90), s2, seed=2)
    r = U.smoothsim_run(model, y, 0.1)
    assert r["why"] != 0


@pytest.mark.parametrize("i", [0, 1, 3, 4, 5, 6])
def test_a_draw_from_the_posterior_equals_the_oracles(i):
    """rand(rng, replace_observation_noise_cov(posterior(model, y), Rn)) with the draws supplied (posterior_lti_sde.jl:48-58 -> lgssm.jl:65-91 on
    the reverse-time model of :193-221): the kernel's recursion with a noise input (k_smooth_one<RAND>, d <= 6) against the oracle's literal loop
    over the EVALUATED posterior"""
    k, dt, s2 = CASES[i]
    T = 5003
    model, y, _ = U.gp_case(k, ("regular", 0.0, dt, T), s2, seed=20 + i)
    d = len(model["x0m"])
    if d > 6:
        pytest.skip("the draw's kernel holds d <= 6")
    rng = np.random.default_rng(i)
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    for Rn in (0.05, rng.random(T) * 0.1):
        post = ref.replace_observation_noise_cov(ref.posterior(model, y), np.broadcast_to(Rn, (T,)).copy())
        want = ref.rand(post, *eps)
        r = U.smoothsim_run(model, y, Rn, eps=eps)
        assert r["rc"] == 0 and r["why"] == 0, r
        assert np.abs(r["mean"] - want).max() <= 1e-8 * max(1.0, np.abs(want).max()), np.abs(r["mean"] - want).max()


def test_a_halo_longer_than_a_span():
    """two Matern-1/2 of one (long) length scale at a short spacing: the recursions forget within 1520 steps -- under the 1536 the tile chain covers,
    but longer than the 1056 steps a span then holds: the second workgroup's run-in would reach back into the head and in front of the series (found
    as a GPU memory fault by scripts/stress_modal.py, seed 101, case 367).  It starts behind the head instead, from the head's exact end state."""
    spec = ("sum", ("scaled", 0.718, ("stretched", 0.6924527157043311, ("matern12",))), ("scaled", 0.605, ("stretched", 0.6924527157043311, ("matern12",))))
    T = 20000
    model, y, _ = U.gp_case(spec, ("regular", 0.0, 0.04, T), 2.56e-2, seed=1)
    Rn = np.random.default_rng(3).random(T) * 0.1
    lp, pm, pv = _reference(model, y, Rn)
    r = U.smoothsim_run(model, y, Rn)
    assert r["halo"] > 1365 and r["nwg"] > 3, r
    _check(r, lp, pm, pv)


@pytest.mark.parametrize("i", [0, 3, 4, 6])
def test_an_emission_offset_per_step(i):
    """a GP with a mean function on a regular grid (lti_sde.jl:118-131): every block shared but the emission offset h_t = m(x_t).  The gains never see
    the offset: the stationary structure holds, the kernel subtracts the offset per step (SmoothCall::hh_t)"""
    k, dt, s2 = CASES[i]
    T = 6000
    model, y, _ = U.gp_case(k, ("regular", 0.0, dt, T), s2, seed=40 + i, mean=("custom", lambda t: np.sin(0.7 * t) + 0.1 * t))
    hh = np.asarray(model["h"], dtype=np.float64)
    assert hh.shape[0] == T
    Rn = np.random.default_rng(i).random(T) * 0.1
    lp, pm, pv = _reference(model, y, Rn)
    r = U.smoothsim_run(model, y, Rn, hh_t=hh)
    _check(r, lp, pm, pv)
