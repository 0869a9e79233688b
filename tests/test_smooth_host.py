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
