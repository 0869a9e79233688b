"""CPU tier: the host plan of the wide-state engine (csrc/tgp_wide.hip through the pure host function tgp_wide_plan of libtgp_hip.so; 8 < d <= 63) --
the covariance half of lgssm.jl:99-165 iterated to its fixed point -- against the discrete algebraic Riccati equation (SciPy) and an independent
fixed-point iteration of the RTS smoother's covariance.  The HIP kernels themselves: tests/test_gpu_wide.py."""
import ctypes

import numpy as np
import pytest
from scipy.linalg import solve_discrete_are

from oracle import components as oc

KERNELS = {
    9: ("product", ("matern52",), ("stretched", 0.7, ("matern52",))),
    12: ("product", ("matern32",), ("approx_periodic", 3, 1.0)),
    28: ("product", ("approx_periodic", 7, 1.0), ("matern32",)),
    42: ("product", ("approx_periodic", 7, 1.0), ("matern52",)),
}


def plan(model, T, post=True):
    from temporalgps_jl_amd import _lib
    lib = _lib.load()
    d = len(model["x0m"])
    c = lambda x: np.ascontiguousarray(np.asarray(x, dtype=np.float64))      # noqa: E731
    A, a, Q = c(model["A"][0].T), c(model["a"][0]), c(model["Q"][0].T)      # column-major blocks
    H, hh, R = c(model["H"][0]), c(np.atleast_1d(model["h"])[:1]), c(np.atleast_1d(model["R"])[:1])
    x0m, x0P = c(model["x0m"]), c(model["x0P"].T)
    info = np.zeros(8, dtype=np.int64)
    K, S, vp = np.zeros(d), np.zeros(1), np.zeros(2)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)      # noqa: E731
    rc = lib.tgp_wide_plan(d, p(A), p(a), p(Q), p(H), p(hh), p(R), p(x0m), p(x0P), T, 1 if post else 0, p(info), p(K), p(S), p(vp))
    assert rc == 0
    return info, K, float(S[0]), vp


@pytest.mark.parametrize("d", sorted(KERNELS))
def test_stationary_gain_solves_the_riccati_equation(d):
    T = 200_000
    model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.1)
    assert len(model["x0m"]) == d
    info, K, S, vp = plan(model, T)
    assert info[0] == 0 and 0 < info[1] < 8192 and info[2] > 0, info
    A, Q, H, R = model["A"][0], model["Q"][0], model["H"][0], float(model["R"][0])
    Pp = solve_discrete_are(A.T, H[:, None], Q, np.array([[R]]))      # the predicted covariance's fixed point
    S_ref = float(H @ Pp @ H + R)
    K_ref = Pp @ H / S_ref
    assert abs(S - S_ref) <= 1e-10 * S_ref, (S, S_ref)
    assert np.max(np.abs(K - K_ref)) <= 1e-9 * max(1.0, np.abs(K_ref).max())
    # the closed loop forgets: |Phi^halo| <= 2^-60
    Phi = (np.eye(d) - np.outer(K, H)) @ A
    assert np.abs(np.linalg.matrix_power(Phi, int(info[2]))).sum(axis=1).max() <= 2.0 ** -59
    # the posterior half: the stationary smoothed emission variance h' Ps h = vbase - qinf, against the RTS smoother's covariance iterated to ITS fixed point
    assert info[3] == 0 and 0 < info[4] <= 8192 and info[5] > 0, info
    Pf = Pp - np.outer(K, K) * S_ref
    G = Pf @ A.T @ np.linalg.inv(Pp)
    Ps = Pf.copy()
    for _ in range(20000):
        Pn = Pf + G @ (Ps - Pp) @ G.T
        if np.max(np.abs(Pn - Ps)) <= 1e-16 * np.abs(Pn).max():
            break
        Ps = Pn
    v_ref = float(H @ Ps @ H)
    assert abs((vp[0] - vp[1]) - v_ref) <= 1e-7 * max(v_ref, 1e-3), (vp, v_ref)      # (the RTS form's own conditioning bounds this check, not the plan's)


def test_plan_declines_what_the_engine_does_not_serve():
    # ApproxPeriodicKernel() alone: no damping, no process noise -- the covariance never settles
    model = oc.build_lgssm(("approx_periodic", 7, 1.0), ("regular", 0.0, 0.1, 100_000), 0.1)
    info, _, _, _ = plan(model, 100_000, post=False)
    assert info[0] != 0, info
    # fewer than 64 steps behind the head
    model = oc.build_lgssm(KERNELS[28], ("regular", 0.0, 0.1, 120), 0.1)
    info, _, _, _ = plan(model, 120, post=False)
    assert info[0] == 4, info
