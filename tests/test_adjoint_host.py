"""CPU tier: the host half of the adjoint gradient (tgp_adjoint_finish, csrc/tgp_adjoint_host.hpp) on a record that NumPy builds
the way the device does (stationary gains behind step n0, sums over the steps behind the head), against central finite differences
of the ORACLE's sequential logpdf with respect to every model block. Also pins the record layout (tgp_steady.hpp: GradRec)."""
import ctypes

import numpy as np
import pytest

from oracle import lgssm_ref as ref
from tests import _util as U

TILE = 512


def device_like_record(model, y, tol=4.5e-16):
    """what k_setup_core / k_apply_grad / k_final_grad leave for an LTI scalar-output model (NumPy, sequential)"""
    A, a, Q, h, hh, R = model["A"][0], model["a"][0], model["Q"][0], model["H"][0], float(model["h"][0]), float(model["R"][0])
    x0m, P = model["x0m"], model["x0P"].copy()
    d, T = len(x0m), len(y)
    kA, S = [], []
    n0 = None
    settled = False
    for t in range(4 * TILE):
        Pp = A @ P @ A.T + Q
        v = Pp @ h
        s = h @ v + R
        kA.append(A @ v / s)
        S.append(s)
        if settled:
            n0 = t
            break
        Pn = Pp - np.outer(v, v) / s
        scale = 0.5 * (np.diag(Pp)[:, None] + np.diag(Pp)[None, :])
        settled = not np.any(np.abs(Pn - P) > tol * scale)
        P = Pn
    assert n0 is not None, "covariance did not settle"
    th = n0 // TILE + 1
    nh = th * TILE
    assert nh + 2 <= T
    ix = lambda t: min(t, n0)
    mu = A @ x0m + a
    mus, rs = np.zeros((T, d)), np.zeros(T)
    for t in range(T):
        mus[t] = mu
        rs[t] = y[t] - hh - h @ mu
        mu = A @ mu + a + kA[ix(t)] * rs[t]
    psi = np.zeros(d)                  # psi_{t+1} behind step t
    SA, Sa, Sk, Srm = np.zeros((d, d)), np.zeros(d), np.zeros(d), np.zeros(d)
    Sr = SSQ = 0.0
    for t in range(T - 1, nh - 1, -1):
        SA += np.outer(psi, mus[t])
        Sa += psi
        Sk += psi * rs[t]
        Srm += rs[t] * mus[t]
        Sr += rs[t]
        SSQ += rs[t] ** 2
        rho = -rs[t] / S[n0] + kA[n0] @ psi
        psi = A.T @ psi - h * rho
    x0P = model["x0P"]
    packed = np.concatenate([x0m, np.array([x0P[r, c] for c in range(d) for r in range(c + 1)])])
    rec = np.concatenate([SA.reshape(-1), Sa, Sk, Srm, [Sr, SSQ], psi, mus[nh], [n0, th, T, 1.0],
                          A.T.reshape(-1), a, Q.T.reshape(-1), h, [hh, R], packed])
    lp = -0.5 * sum(np.log(2 * np.pi) + np.log(S[ix(t)]) + rs[t] ** 2 / S[ix(t)] for t in range(T))
    return np.ascontiguousarray(rec), nh, lp


def finish(lib, d, rec, y_head):
    out = dict(A=np.zeros((d, d)), a=np.zeros(d), Q=np.zeros((d, d)), H=np.zeros(d), h=np.zeros(1), R=np.zeros(1), x0m=np.zeros(d), x0P=np.zeros((d, d)))
    p = lambda x: x.ctypes.data
    rc = lib.tgp_adjoint_finish(d, p(rec), p(y_head), len(y_head), *[p(out[k]) for k in ("A", "a", "Q", "H", "h", "R", "x0m", "x0P")])
    assert rc == 0
    for k in ("A", "Q", "x0P"):
        out[k] = out[k].T.copy()       # column-major -> [i][k]
    return out


@pytest.mark.parametrize("d", [1, 2, 3, 5])
def test_host_half_against_finite_differences_of_the_oracle(d):
    import temporalgps_jl_amd as tgp
    lib = tgp._lib.load()
    assert lib.tgp_adjoint_record_size(d) == 3 * d * d + 8 * d + 8 + d * (d + 1) // 2
    rng = np.random.default_rng(40 + d)
    T = 700
    model = U.random_lgssm(rng, False, d, T)
    y = ref.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    rec, nh, lp_engine = device_like_record(model, y)
    assert len(rec) == lib.tgp_adjoint_record_size(d)
    lp = ref.logpdf(model, y)
    assert abs(lp_engine - lp) <= 1e-11 * abs(lp)
    g = finish(lib, d, rec, np.ascontiguousarray(y[:nh]))

    def fd(key, idx, sym=False, step=1e-6):
        vals = []
        for sgn in (1.0, -1.0):
            m = {k: (np.array(v, dtype=float, copy=True) if isinstance(v, np.ndarray) else v) for k, v in model.items()}
            arr = m[key]
            tgt = arr[0] if key in ("A", "a", "Q", "H", "h", "R") else arr
            if key in ("h", "R"):
                arr[0] += sgn * step
            else:
                tgt[idx] += sgn * step
                if sym and idx[0] != idx[1]:
                    tgt[idx[::-1]] += sgn * step
            vals.append(ref.logpdf(m, y))
        return (vals[0] - vals[1]) / (2 * step)

    scale = max(1.0, max(np.abs(v).max() for v in g.values()))
    tol = 2e-6 * scale
    for i in range(d):
        assert abs(fd("a", (i,)) - g["a"][i]) <= tol
        assert abs(fd("H", (i,)) - g["H"][i]) <= tol
        assert abs(fd("x0m", (i,)) - g["x0m"][i]) <= tol
        for k in range(d):
            assert abs(fd("A", (i, k)) - g["A"][i, k]) <= tol, ("A", i, k)
            # symmetric blocks: a symmetric perturbation E_ik + E_ki pairs with g_ik + g_ki (g symmetrised)
            w = 1.0 if i == k else 2.0
            assert abs(fd("Q", (i, k), sym=True) - w * g["Q"][i, k]) <= tol, ("Q", i, k)
            assert abs(fd("x0P", (i, k), sym=True) - w * g["x0P"][i, k]) <= tol, ("x0P", i, k)
    assert abs(fd("h", None) - g["h"][0]) <= tol
    assert abs(fd("R", None) - g["R"][0]) <= tol


def test_bad_records_are_refused():
    import temporalgps_jl_amd as tgp
    lib = tgp._lib.load()
    d = 2
    rec = np.zeros(lib.tgp_adjoint_record_size(d))
    yh = np.zeros(TILE)
    z = np.zeros(16)
    p = lambda x: x.ctypes.data
    assert lib.tgp_adjoint_finish(d, p(rec), p(yh), TILE, *[p(z)] * 8) == tgp._lib.EINVAL       # "applies" flag is 0
    assert lib.tgp_adjoint_finish(9, p(rec), p(yh), TILE, *[p(z)] * 8) == tgp._lib.EINVAL
    assert lib.tgp_adjoint_record_size(0) == 0 and lib.tgp_adjoint_record_size(9) == 0
