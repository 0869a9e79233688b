"""ORACLE (test infrastructure only). ctypes wrapper around oracle/liboracle_seq.so
(C restatement, see seq_kalman.c). Models use the dict convention of oracle/lgssm_ref.py,
kind == 'scalar', ordering == 'F', d <= 8."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_dp = ctypes.POINTER(ctypes.c_double)
_i64 = ctypes.c_int64


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_seq.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(arr):
    return None if arr is None else arr.ctypes.data_as(_dp)


def _blocks(model):
    """C-contiguous copies in Julia (column-major block) layout + per-step strides (0 == Fill)."""
    d = len(model["x0m"])
    A = np.ascontiguousarray(np.swapaxes(model["A"], -1, -2))     # row-major of A' == col-major of A
    Q = np.ascontiguousarray(np.swapaxes(model["Q"], -1, -2))
    a = np.ascontiguousarray(model["a"])
    H = np.ascontiguousarray(model["H"])
    h = np.ascontiguousarray(np.atleast_1d(model["h"]))
    R = np.ascontiguousarray(np.atleast_1d(model["R"]))
    st = lambda arr, n: _i64(n if arr.shape[0] > 1 else 0)
    x0P = np.ascontiguousarray(model["x0P"].T)
    x0m = np.ascontiguousarray(model["x0m"])
    args = [_p(A), st(A, d * d), _p(a), st(a, d), _p(Q), st(Q, d * d), _p(H), st(H, d),
            _p(h), st(h, 1), _p(R), st(R, 1)]
    keep = (A, Q, a, H, h, R, x0m, x0P)
    return d, args, x0m, x0P, keep


def filter_(model, y, want_states=False):
    assert model["kind"] == "scalar" and model["ordering"] == "F"
    d, args, x0m, x0P, keep = _blocks(model)
    T = model["T"]
    y = np.ascontiguousarray(y, dtype=np.float64)
    lml = ctypes.c_double(0.0)
    m_out = np.empty((T, d)) if want_states else None
    P_out = np.empty((T, d, d)) if want_states else None
    rc = lib().oracle_seq_filter(d, _i64(T), *args, _p(y), _p(x0m), _p(x0P), ctypes.byref(lml),
                                 _p(m_out), _p(P_out))
    assert rc == 0, rc
    if want_states:
        return lml.value, m_out, np.swapaxes(P_out, -1, -2).copy()
    return lml.value


def logpdf(model, y):
    return filter_(model, y)


def posterior(model, y):
    d, args, x0m, x0P, keep = _blocks(model)
    T = model["T"]
    y = np.ascontiguousarray(y, dtype=np.float64)
    G, g, L = np.empty((T, d, d)), np.empty((T, d)), np.empty((T, d, d))
    xfm, xfP = np.empty(d), np.empty((d, d))
    rc = lib().oracle_seq_posterior(d, _i64(T), *args, _p(y), _p(x0m), _p(x0P), _p(G), _p(g), _p(L),
                                    _p(xfm), _p(xfP))
    assert rc == 0, rc
    out = dict(model)
    out.update(ordering="R", A=np.swapaxes(G, -1, -2).copy(), a=g, Q=np.swapaxes(L, -1, -2).copy(),
               x0m=xfm, x0P=xfP.T.copy())
    return out


def posterior_marginals(model, y, R_new):
    d, args, x0m, x0P, keep = _blocks(model)
    T = model["T"]
    y = np.ascontiguousarray(y, dtype=np.float64)
    Rn = np.ascontiguousarray(np.atleast_1d(R_new), dtype=np.float64)
    G, g, L = np.empty((T, d, d)), np.empty((T, d)), np.empty((T, d, d))
    mean, var = np.empty(T), np.empty(T)
    rc = lib().oracle_seq_posterior_marginals(
        d, _i64(T), *args, _p(y), _p(x0m), _p(x0P), _p(Rn), _i64(1 if Rn.shape[0] > 1 else 0),
        _p(G), _p(g), _p(L), _p(mean), _p(var))
    assert rc == 0, rc
    return mean, var


def prior_marginals(model):
    d, args, x0m, x0P, keep = _blocks(model)
    T = model["T"]
    mean, var = np.empty(T), np.empty(T)
    rc = lib().oracle_seq_prior_marginals(d, _i64(T), *args, _p(x0m), _p(x0P), _p(mean), _p(var))
    assert rc == 0, rc
    return mean, var


def rand(model, eps_t, eps_e, eps_0):
    d, args, x0m, x0P, keep = _blocks(model)
    T = model["T"]
    et = np.ascontiguousarray(eps_t, dtype=np.float64)
    ee = np.ascontiguousarray(eps_e, dtype=np.float64)
    e0 = np.ascontiguousarray(eps_0, dtype=np.float64)
    y = np.empty(T)
    rc = lib().oracle_seq_rand(d, _i64(T), *args, _p(x0m), _p(x0P), _p(et), _p(ee), _p(e0), _p(y))
    assert rc == 0, rc
    return y
