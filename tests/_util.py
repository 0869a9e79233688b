"""Shared helpers for the test tiers (test infrastructure)."""
import ctypes
import os
import subprocess

import numpy as np

from oracle import components as oc
from oracle import lgssm_ref as ref

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_dp = ctypes.POINTER(ctypes.c_double)
_i64 = ctypes.c_int64


def pack(model):
    """dict model (oracle convention; kind 'scalar', or 'small' with DIAGONAL R) -> flat blocks + strides (0 == Fill)."""
    d = len(model["x0m"])
    if model["kind"] == "small":
        p_ = model["H"].shape[-2]
        R = np.atleast_3d(model["R"])
        assert np.allclose(R, R * np.eye(p_)), "pack(): dense R must be whitened first"
        model = dict(model, R=np.diagonal(R, axis1=-2, axis2=-1))
    A = np.ascontiguousarray(np.swapaxes(model["A"], -1, -2)).reshape(-1)
    Q = np.ascontiguousarray(np.swapaxes(model["Q"], -1, -2)).reshape(-1)
    a = np.ascontiguousarray(model["a"]).reshape(-1)
    H = np.ascontiguousarray(model["H"]).reshape(-1)
    h = np.ascontiguousarray(np.atleast_1d(model["h"]), dtype=np.float64).reshape(-1)
    R = np.ascontiguousarray(np.atleast_1d(model["R"]), dtype=np.float64).reshape(-1)
    st = lambda arr, n: n if arr.shape[0] > 1 else 0
    p = model["H"].shape[-2] if model["kind"] == "small" else 1
    return dict(d=d, p=p, small=int(model["kind"] == "small"), T=model["T"], ordering=0 if model["ordering"] == "F" else 1,
                A=A, sA=st(model["A"], d * d), a=a, sa=st(model["a"], d), Q=Q, sQ=st(model["Q"], d * d),
                H=H, sH=st(model["H"], p * d), h=h, sh=st(np.atleast_1d(model["h"]), p),
                R=R, sR=st(np.atleast_1d(model["R"]), p),
                x0m=np.ascontiguousarray(model["x0m"], dtype=np.float64),
                x0P=np.ascontiguousarray(model["x0P"].T, dtype=np.float64).reshape(-1))


def is_lti(pk):
    return pk["sA"] == 0 and pk["sa"] == 0 and pk["sQ"] == 0 and pk["sH"] == 0 and pk["sh"] == 0


def gp_case(k, t, s2, seed, mean=None):
    """Model from a kernel spec + a draw y from it (as the reference bench does, single_output_gps.jl:143-145)."""
    rng = np.random.default_rng(seed)
    model = oc.build_lgssm(k, t, s2, mean)
    T, d = model["T"], len(model["x0m"])
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    return model, ref.rand(model, *eps), eps


def random_lgssm(rng, tv, d, T, ordering="F", tame_reverse=False):
    """test/models/model_test_utils.jl:163-263 (scalar-output variants), stable transitions."""
    def psd(n, lo, hi):
        U = np.linalg.qr(rng.standard_normal((n, n)))[0]
        return (U * (rng.random(n) * (hi - lo) + lo)) @ U.T
    x0m, x0P = rng.standard_normal(d), psd(d, 0.9, 1.1)
    n = T if tv else 1
    A = np.stack([-psd(d, 0.1, 0.9) + 0.2 * rng.standard_normal((d, d)) for _ in range(n)])
    A = np.stack([Ai / max(1.0, 1.1 * np.abs(np.linalg.eigvals(Ai)).max()) for Ai in A])
    a = rng.standard_normal((n, d))
    Q = np.stack([psd(d, 0.2, 1.5) for _ in range(n)])
    H, h, R = rng.standard_normal((n, d)), rng.standard_normal(n), rng.random(n) + 0.1
    if ordering == "R" and tame_reverse:
        # (asked for by the one test that runs the POSTERIOR of a Reverse model forward; the other Reverse tests keep the plain draw)
        # step_posterior(::Reverse) (lgssm.jl:223-228) calls invert_dynamics with predicted and filtered state swapped: its G is
        # (A Pf A' + Q) A' Pf^-1, contractive only for weak transitions and weakly informative observations. Models are drawn there, so
        # that the posterior of a Reverse model can be run forward over hundreds of steps without overflow (in the oracle as well).
        A = 0.3 * A
        R = R + 2.0
    return dict(ordering=ordering, kind="scalar", T=T, A=A, a=a, Q=Q, H=H, h=h, R=R, x0m=x0m, x0P=x0P)


# --------------------------------------------------------------------------- hostsim (CPU emulation of the chunk algorithm)
_HOSTSIM = None


def hostsim():
    global _HOSTSIM
    if _HOSTSIM is None:
        src = os.path.join(HERE, "hostsim", "hostsim.cpp")
        so = os.path.join(HERE, "hostsim", "libhostsim.so")
        deps = [src] + [os.path.join(ROOT, "temporalgps.jl_amd", "csrc", f) for f in ("tgp_math.hpp", "tgp_chunk.hpp", "tgp_math_body.inc", "tgp_chunk_body.inc")]
        if not os.path.exists(so) or any(os.path.getmtime(p) > os.path.getmtime(so) for p in deps):
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so, src])
        _HOSTSIM = ctypes.CDLL(so)
    return _HOSTSIM


def _p(x):
    return None if x is None else x.ctypes.data_as(_dp)


def hostsim_run(model, what, y=None, missing=None, L0=4, BS=3, Rnew=None, eps=None, want_ggl=False, want_elem=False,
                want_rev=False, xs=None):
    pk = pack(model)
    d, T, p_ = pk["d"], pk["T"], pk["p"]
    L0 = ((L0 + p_ - 1) // p_) * p_          # whole time steps per chunk
    osh = (T,) if p_ == 1 else (T, p_)
    out = {}
    lml = ctypes.c_double(0.0)
    m_out = P_out = G = g = L = xfm = xfP = mean = var = None
    if what == 1:
        m_out, P_out = np.zeros((T, d)), np.zeros((T, d, d))
    if what == 2:
        xfm, xfP = np.zeros(d), np.zeros((d, d))
        if want_ggl:
            G, g, L = np.zeros((T, d, d)), np.zeros((T, d)), np.zeros((T, d, d))
        if Rnew is not None:
            mean, var = np.zeros(osh), np.zeros(osh)
    if what in (3, 4):
        mean, var = np.zeros(osh), np.zeros(osh)
    Rn = None if Rnew is None else np.ascontiguousarray(np.atleast_1d(Rnew), dtype=np.float64)
    miss = None if missing is None else np.ascontiguousarray(missing, dtype=np.uint8)
    yv = np.zeros(osh) if y is None else np.ascontiguousarray(y, dtype=np.float64)
    et = ee = None
    x0m = pk["x0m"]
    x0P = pk["x0P"]
    if what == 4:
        et = np.ascontiguousarray(eps[0], dtype=np.float64)
        ee = np.ascontiguousarray(eps[1], dtype=np.float64)
        x0m = ref.rand_x0(eps[2], model["x0m"], model["x0P"])
        x0P = np.zeros(d * d)
    d_ = pk["d"]
    elem = np.zeros(d_ * d_ + 2 * d_ + d_ * (d_ + 1)) if want_elem else None
    rev = np.zeros(d_ * d_ + d_ + d_ * (d_ + 1) // 2) if want_rev else None
    xs_m = None if xs is None else np.ascontiguousarray(xs[0], dtype=np.float64)
    xs_P = None if xs is None else np.ascontiguousarray(np.asarray(xs[1]).T, dtype=np.float64)
    rc = hostsim().hostsim_run(
        d, p_, pk["small"], int(is_lti(pk)), what, L0, BS, _i64(T), pk["ordering"], _p(pk["A"]), _i64(pk["sA"]), _p(pk["a"]), _i64(pk["sa"]),
        _p(pk["Q"]), _i64(pk["sQ"]), _p(pk["H"]), _i64(pk["sH"]), _p(pk["h"]), _i64(pk["sh"]), _p(pk["R"]), _i64(pk["sR"]),
        _p(yv), None if miss is None else miss.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
        _p(x0m), _p(x0P), ctypes.byref(lml), _p(m_out), _p(P_out), _p(G), _p(g), _p(L), _p(xfm), _p(xfP),
        _p(Rn), _i64(0 if Rn is None or Rn.size == p_ else 1), _p(mean), _p(var), _p(et), _p(ee),
        _p(elem), _p(rev), _p(xs_m), _p(xs_P))
    out.update(rc=rc, lml=lml.value, m=m_out, P=None if P_out is None else np.swapaxes(P_out, -1, -2),
               G=None if G is None else np.swapaxes(G, -1, -2), g=g, L=None if L is None else np.swapaxes(L, -1, -2),
               xfm=xfm, xfP=None if xfP is None else xfP.T, mean=mean, var=var, elem=elem, rev=rev)
    return out


def random_lgssm_small(rng, tv, d, p, T, ordering="F", dense_R=False):
    """SmallOutputLGC emissions (vector observations), test/models/model_test_utils.jl:187-230."""
    m = random_lgssm(rng, tv, d, T, ordering)
    n = T if tv else 1

    def psd(k, lo, hi):
        U_ = np.linalg.qr(rng.standard_normal((k, k)))[0]
        return (U_ * (rng.random(k) * (hi - lo) + lo)) @ U_.T
    m["kind"] = "small"
    m["H"] = rng.standard_normal((n, p, d))
    m["h"] = rng.standard_normal((n, p))
    if dense_R:
        m["R"] = np.stack([psd(p, 0.9, 1.1) for _ in range(n)])
    else:
        m["R"] = np.stack([np.diag(rng.random(p) + 0.1) for _ in range(n)])
    return m


def hostsim_grad(model, dmodel, y, L0=5, BS=3, missing=None):
    """d logpdf / d theta for an LTI scalar model: `dmodel` holds the tangents of A, a, Q, H, h, R, x0m, x0P (same shapes)."""
    pk, dk = pack(model), pack(dict(model, **dmodel))
    assert is_lti(pk) and pk["sR"] == 0
    lml, dl = ctypes.c_double(), ctypes.c_double()
    miss = None if missing is None else np.ascontiguousarray(missing, dtype=np.uint8)
    yv = np.ascontiguousarray(y, dtype=np.float64)
    rc = hostsim().hostsim_grad(pk["d"], L0, BS, _i64(pk["T"]), _p(pk["A"]), _p(pk["a"]), _p(pk["Q"]), _p(pk["H"]), _p(pk["h"]), _p(pk["R"]),
                                _p(yv), None if miss is None else miss.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                _p(pk["x0m"]), _p(pk["x0P"]), _p(dk["A"]), _p(dk["a"]), _p(dk["Q"]), _p(dk["H"]), _p(dk["h"]), _p(dk["R"]),
                                _p(dk["x0m"]), _p(dk["x0P"]), ctypes.byref(lml), ctypes.byref(dl))
    assert rc == 0, rc
    return lml.value, dl.value


def model_tangent(build, theta, k, rel=1e-6):
    """central finite difference of the (tiny, shared) model blocks w.r.t. theta[k]: d(A, a, Q, H, h, R, x0m, x0P)/d theta_k."""
    th = np.array(theta, dtype=np.float64)
    hstep = rel * max(1.0, abs(th[k]))
    tp, tm = th.copy(), th.copy()
    tp[k] += hstep
    tm[k] -= hstep
    mp, mm = build(tp), build(tm)
    return {key: (np.asarray(mp[key], dtype=np.float64) - np.asarray(mm[key], dtype=np.float64)) / (2 * hstep)
            for key in ("A", "a", "Q", "H", "h", "R", "x0m", "x0P")}


# --------------------------------------------------------------------------- sweepsim (CPU emulation of the sweep engine, tgp_sweep.hpp)
_SWEEPSIM = None


def sweepsim():
    global _SWEEPSIM
    if _SWEEPSIM is None:
        src = os.path.join(HERE, "hostsim", "sweepsim.cpp")
        so = os.path.join(HERE, "hostsim", "libsweepsim.so")
        deps = [src] + [os.path.join(ROOT, "temporalgps.jl_amd", "csrc", f) for f in ("tgp_math.hpp", "tgp_sweep_body.hpp", "tgp_sweep_plan.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(p) > os.path.getmtime(so) for p in deps):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so, src])
        _SWEEPSIM = ctypes.CDLL(so)
        _SWEEPSIM.sweepsim_run.restype = ctypes.c_int
    return _SWEEPSIM


_SMOOTHSIM = None


def smoothsim():
    global _SMOOTHSIM
    if _SMOOTHSIM is None:
        src = os.path.join(HERE, "hostsim", "smoothsim.cpp")
        so = os.path.join(HERE, "hostsim", "libsmoothsim.so")
        deps = [src, os.path.join(ROOT, "temporalgps.jl_amd", "csrc", "tgp_steady_plan.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(p) > os.path.getmtime(so) for p in deps):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so, src])
        _SMOOTHSIM = ctypes.CDLL(so)
        _SMOOTHSIM.smoothsim_run.restype = ctypes.c_int
    return _SMOOTHSIM


def smoothsim_run(model, y, Rnew, eps=None, hh_t=None):
    """model: oracle dict with SHARED blocks and one noise variance (an LTI model).  The dense-powers one-launch smoother on the host
    (tests/hostsim/smoothsim.cpp).  Returns dict(rc, lml, why, n0, nhs, n1, halo, nwg, mean, var).
    eps = (eps_t (T, d), eps_e (T,), eps_0 (d,)): a DRAW from the posterior instead (returned as `mean`)."""
    T, d = model["T"], len(model["x0m"])
    cm = lambda M: np.ascontiguousarray(np.asarray(M, dtype=np.float64).T).reshape(-1)
    A, Q = cm(model["A"][0]), cm(model["Q"][0])
    a = np.ascontiguousarray(model["a"][0], dtype=np.float64)
    H = np.ascontiguousarray(model["H"][0], dtype=np.float64)
    x0m = np.ascontiguousarray(model["x0m"], dtype=np.float64)
    x0P = cm(model["x0P"])
    yv = np.ascontiguousarray(y, dtype=np.float64)
    rn = np.ascontiguousarray(np.atleast_1d(Rnew), dtype=np.float64)
    mean, var, out = np.zeros(T), np.zeros(T), np.zeros(8)
    keep = [np.ascontiguousarray(e, dtype=np.float64) for e in eps] if eps is not None else []
    ee = [_p(k) for k in keep] if keep else [None, None, None]
    hk = None if hh_t is None else np.ascontiguousarray(hh_t, dtype=np.float64)      # an emission offset per step (the model dict's h may then be per step too)
    rc = smoothsim().smoothsim_run(d, _p(A), _p(a), _p(Q), _p(H), ctypes.c_double(float(np.atleast_1d(model["h"])[0])),
                                   ctypes.c_double(float(np.atleast_1d(model["R"])[0])), _p(x0m), _p(x0P), _i64(T), _p(yv), _p(rn),
                                   int(rn.shape[0] > 1), _p(mean), _p(var), _p(out), *ee, _p(hk))
    return dict(rc=rc, lml=out[0], why=int(out[1]), n0=int(out[2]), nhs=int(out[3]), n1=int(out[4]), halo=int(out[5]), nwg=int(out[6]),
                mean=mean, var=var)


def kernel_sde(k):
    """(F, H) of a kernel expression built from Matern terms with scaled / stretched / sum (what the closed-form transitions cover)."""
    from scipy.linalg import block_diag
    if k[0] == "sum":
        parts = [kernel_sde(kk) for kk in k[1:]]
        return block_diag(*[p[0] for p in parts]), np.concatenate([p[1] for p in parts])
    if k[0] == "scaled":
        F, H = kernel_sde(k[2])
        return F, np.sqrt(k[1]) * H
    if k[0] == "stretched":
        F, H = kernel_sde(k[2])
        return F * k[1], H
    F, _, H = oc.to_sde(k)
    return F, H


def sde_coef(F):
    """[lambda per row | N | N^2 / 2] of a block-diagonal drift with one eigenvalue per block (ModelView::sde; tgp_api.hip sde_closed_form)."""
    d = F.shape[0]
    lam, N, N2 = np.zeros(d), np.zeros((d, d)), np.zeros((d, d))
    seen, i = set(), 0
    while i < d:
        S = [i]
        grow = True
        while grow:
            grow = False
            for j in range(d):
                if j not in S and any(F[j, s] != 0 or F[s, j] != 0 for s in S):
                    S.append(j)
                    grow = True
        S = sorted(S)
        n = len(S)
        l = -np.trace(F[np.ix_(S, S)]) / n
        Nb = F[np.ix_(S, S)] + l * np.eye(n)
        assert np.allclose(np.linalg.matrix_power(Nb, n), 0.0, atol=1e-12 * max(1.0, np.abs(F).max()) ** n)
        lam[S] = l
        if n >= 2:
            N[np.ix_(S, S)] = Nb
        if n >= 3:
            N2[np.ix_(S, S)] = 0.5 * Nb @ Nb
        seen.update(S)
        i = max(S) + 1
    return np.concatenate([lam, N.T.reshape(-1), N2.T.reshape(-1)])     # column-major blocks


def sweepsim_run(model, y, missing=None, Rnew=None, post=True, sde=None, C=0, W=0, Wb=0, num_cu=1, w_hint=0, wb_hint=0):
    """model: oracle dict with SHARED A, a, Q, H (or, with sde = (F, times): transitions by the closed form from the gaps); R / h may be
    per step.  Returns dict(rc, lml, status, dist_f, dist_b, C, W, Wb, nwaves, mean, var)."""
    T, d = model["T"], len(model["x0m"])
    cm = lambda M: np.ascontiguousarray(np.asarray(M, dtype=np.float64).T).reshape(-1)
    R = np.atleast_1d(np.asarray(model["R"], dtype=np.float64))
    h = np.atleast_1d(np.asarray(model["h"], dtype=np.float64))
    Rstep = np.ascontiguousarray(R) if R.shape[0] > 1 else None
    hstep = np.ascontiguousarray(h) if h.shape[0] > 1 else None
    Rrep = float(np.median(R[R < 1e14])) if np.any(R < 1e14) else 1.0
    coef, tau, tau_typ = None, None, 0.0
    if sde is not None:
        F, times_ = sde
        coef = sde_coef(F)
        tau = np.concatenate([[-1.0], np.diff(np.asarray(times_, dtype=np.float64))])
        tau_typ = float(np.median(tau[1:])) if T > 1 else 1.0
    A, Q = cm(model["A"][0]), cm(model["Q"][0])
    a = np.ascontiguousarray(model["a"][0], dtype=np.float64)
    H = np.ascontiguousarray(model["H"][0], dtype=np.float64)
    x0m = np.ascontiguousarray(model["x0m"], dtype=np.float64)
    x0P = cm(model["x0P"])
    yv = np.ascontiguousarray(y, dtype=np.float64)
    mk = None if missing is None else np.ascontiguousarray(missing, dtype=np.uint8)
    rn = np.ascontiguousarray(np.atleast_1d(Rnew if Rnew is not None else 0.0), dtype=np.float64)
    mean, var, out = np.zeros(T), np.zeros(T), np.zeros(8)
    u8 = ctypes.POINTER(ctypes.c_uint8)
    rc = sweepsim().sweepsim_run(
        d, int(sde is not None), _i64(T), _p(A), _p(a), _p(Q), _p(H), ctypes.c_double(float(h[0])), ctypes.c_double(Rrep), _p(x0m), _p(x0P),
        _p(coef), ctypes.c_double(tau_typ), _p(yv), None if mk is None else mk.ctypes.data_as(u8), _p(Rstep), _p(hstep), _p(tau), _p(rn),
        int(rn.shape[0] > 1), int(post), C, W, Wb, w_hint, wb_hint, num_cu, _p(mean), _p(var), _p(out))
    return dict(rc=rc, lml=out[0], status=int(out[1]), dist_f=out[2], dist_b=out[3], C=int(out[4]), W=int(out[5]), Wb=int(out[6]),
                nwaves=int(out[7]), mean=mean, var=var)
