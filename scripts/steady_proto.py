"""NumPy prototype of the stationary-gain scan engine (csrc/tgp_steady.hip) -- a development aid, not product code.

Mirrors the kernels' structure one to one (setup tables -> tile elements -> carries -> apply) so that index
conventions can be checked on the CPU against oracle/lgssm_ref.py before / while the HIP code is debugged on
the GPU box.  Run:  python scripts/steady_proto.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TILE = 512
SUB = 8
LANES = 64
LOG2PI = float(np.log(2.0 * np.pi))


def sym_u(P):
    U = np.triu(P)
    return U + np.triu(P, 1).T


def setup(A, a, Q, h, hh, R, x0m, x0P, T, head_max=4096, tol=4.5e-16):
    """Data-independent tables (k_steady_setup).  Returns a dict or None when the path does not apply."""
    d = A.shape[0]
    P = sym_u(x0P).copy()
    ent = []                     # per step: kA, rS, iS, logS, Pf, Pp, K
    Pold2 = None
    tc = None
    t = 0
    while True:
        Pf = P
        Pp = (A @ sym_u(Pf)) @ A.T + Q
        V = h @ Pp
        S = V @ h + R
        if not S > 0:
            return None
        K = V / S
        rs = 1.0 / np.sqrt(S)
        B = V * rs
        Pn = Pp - np.outer(B, B)
        ent.append(dict(kA=A @ K, rS=R / S, iS=1.0 / S, logS=np.log(S), Pf=Pf, Pp=Pp, K=K))
        if tc is not None:
            break                # this was the extra (steady) iteration
        dg = np.diag(Pp)
        scale = 0.5 * (dg[:, None] + dg[None, :])
        conv = not np.any(np.abs(Pn - Pf) > tol * scale)
        cyc = Pold2 is not None and np.array_equal(Pn, Pold2)
        Pold2 = Pf
        P = Pn
        if conv or cyc:
            tc = t
        t += 1
        if t > head_max:
            return None
    n0 = tc + 1                  # entries 0..n0 exist; entry n0 is the steady one
    th = n0 // TILE + 1
    nh = th * TILE
    Pss = P
    # (b) reverse-time gains of the head steps
    for t in range(0, n0 + 1):
        e = ent[t]
        U = np.linalg.cholesky(sym_u(e["Pp"]) + 1e-10 * np.eye(d)).T
        Gt = np.linalg.solve(U, np.linalg.solve(U.T, A @ e["Pf"]))
        e["G"] = Gt.T
        W = U @ Gt
        e["L"] = e["Pf"] - W.T @ W
        e["c"] = e["G"] @ e["K"]
    ss = ent[n0]
    G, L = ss["G"], ss["L"]
    # (c) tail of the smoothed covariance
    Ps = Pss.copy()
    vbt = []
    Pold2 = None
    n1 = None
    for j in range(head_max):
        vbt.append(h @ sym_u(Ps) @ h)
        Pn = (G @ sym_u(Ps)) @ G.T + L
        dg = np.abs(np.diag(Ps)) + np.abs(np.diag(Pn))
        scale = 0.25 * (dg[:, None] + dg[None, :])
        conv = not np.any(np.abs(Pn - Ps) > tol * scale)
        cyc = Pold2 is not None and np.array_equal(Pn, Pold2)
        Pold2 = Ps
        Ps = Pn
        if conv or cyc:
            n1 = j + 1
            break
    if n1 is None or nh + n1 + 1 > T:
        return None
    Psss = Ps
    vb_ss = h @ sym_u(Psss) @ h
    # (d) head of the smoothed covariance
    vbh = np.full(nh, vb_ss)
    Ps = Psss
    for t in range(n0, 0, -1):
        vbh[t] = h @ sym_u(Ps) @ h if t < nh else vbh[t - 1]
        Ps = (ent[t]["G"] @ sym_u(Ps)) @ ent[t]["G"].T + ent[t]["L"]
    vbh[0] = h @ sym_u(Ps) @ h
    # (e) powers
    Phi = A - np.outer(ss["kA"], h)
    kp = int(np.ceil(np.log2(max(T, 2)))) + 2
    PhiPow = [Phi]
    GPow = [G]
    Bm = np.outer(ss["c"], h)
    ntiles = (T + TILE - 1) // TILE
    nv = T - (ntiles - 1) * TILE          # valid steps of the last tile: its coupling is B_nv, composed from the B_(2^k)
    Bl = np.zeros((d, d)); Xacc = np.eye(d); Gacc = np.eye(d)
    for k in range(kp):
        if k < 9 and (nv >> k) & 1:
            Bl = Bl + Gacc @ Bm @ Xacc
            Xacc = PhiPow[k] @ Xacc
            Gacc = GPow[k] @ Gacc
        if k < 9:
            Bm = Bm + GPow[k] @ Bm @ PhiPow[k]
        PhiPow.append(PhiPow[k] @ PhiPow[k])
        GPow.append(GPow[k] @ GPow[k])
    tab = dict(d=d, n0=n0, th=th, nh=nh, n1=n1, ss=ss, vb_ss=vb_ss, vbt=np.array(vbt), vbh=vbh,
               PhiPow=PhiPow, GPow=GPow, B=Bm, Bl=(Bm if nv == TILE else Bl), A=A, a=a, h=h, hh=hh, R=R,
               LS_head=sum(e["logS"] for e in ent[:n0]))
    # head tables filled up to nh with the steady entry
    def col(name):
        return np.array([ent[min(t, n0)][name] for t in range(nh)])
    tab["kA_t"] = col("kA"); tab["rS_t"] = col("rS"); tab["iS_t"] = col("iS")
    tab["G_t"] = col("G"); tab["c_t"] = col("c")
    tab["mu0"] = A @ x0m + a
    return tab


def _wave_scan_const(f, Mpow):
    """Inclusive Kogge-Stone scan over 64 lanes of s_l = M s_{l-1} + f_l with M^(2^k) = Mpow[k]."""
    f = f.copy()
    for k in range(6):
        off = 1 << k
        g = np.zeros_like(f)
        g[off:] = f[:-off]
        f[off:] = f[off:] + g[off:] @ Mpow[k].T
    return f


def _wave_scan_rev_const(b, Mpow):
    b = b.copy()
    for k in range(6):
        off = 1 << k
        g = np.zeros_like(b)
        g[:-off] = b[off:]
        b[:-off] = b[:-off] + g[:-off] @ Mpow[k].T
    return b


def tile_steady(tab, y8, valid, mu_in, lam_in, want_out):
    """One steady tile.  y8, valid: [64][8].  mu_in: carry into lane 0 (d) ; lam_in: carry into lane 63.
    Returns f_tile, b_tile, sum r^2, (mean8)"""
    d = tab["d"]; A = tab["A"]; a = tab["a"]; h = tab["h"]; hh = tab["hh"]; ss = tab["ss"]
    kA = ss["kA"]; G = ss["G"]; c = ss["c"]
    mu = np.zeros((LANES, d)); mu[0] = mu_in
    for j in range(SUB):
        r = np.where(valid[:, j], y8[:, j] - hh - mu @ h, 0.0)
        mu = mu @ A.T + a + np.outer(r, kA)
    f = _wave_scan_const(mu, tab["PhiPow"][3:9])
    start = np.zeros((LANES, d)); start[1:] = f[:-1]; start[0] = mu_in
    mu = start.copy()
    r8 = np.zeros((LANES, SUB))
    for j in range(SUB):
        r = np.where(valid[:, j], y8[:, j] - hh - mu @ h, 0.0)
        r8[:, j] = r
        mu = mu @ A.T + a + np.outer(r, kA)
    lam = np.zeros((LANES, d)); lam[LANES - 1] = lam_in
    for j in range(SUB - 1, -1, -1):
        lam = lam @ G.T + np.outer(r8[:, j], c)
    b = _wave_scan_rev_const(lam, tab["GPow"][3:9])
    out = None
    if want_out:
        lstart = np.zeros((LANES, d)); lstart[:-1] = b[1:]; lstart[LANES - 1] = lam_in
        lam = lstart
        out = np.zeros((LANES, SUB))
        for j in range(SUB - 1, -1, -1):
            out[:, j] = y8[:, j] - ss["rS"] * r8[:, j] + lam @ h
            lam = lam @ G.T + np.outer(r8[:, j], c)
    return f[LANES - 1], b[0], np.sum(r8 * r8), out


def run(model, y, Rnew):
    A = np.asarray(model["A"], float)[0]; a = np.asarray(model["a"], float)[0]; Q = np.asarray(model["Q"], float)[0]
    h = np.asarray(model["H"], float).reshape(-1); hh = float(np.asarray(model["h"]).reshape(-1)[0])
    R = float(np.asarray(model["R"]).reshape(-1)[0])
    T = len(y)
    d = A.shape[0]
    tab = setup(A, a, Q, h, hh, R, np.asarray(model["x0m"], float), np.asarray(model["x0P"], float), T)
    if tab is None:
        return None
    nh, th, n0, n1 = tab["nh"], tab["th"], tab["n0"], tab["n1"]
    ntiles = (T + TILE - 1) // TILE
    ypad = np.zeros(ntiles * TILE); ypad[:T] = y
    vpad = np.zeros(ntiles * TILE, bool); vpad[:T] = True
    y3 = ypad.reshape(ntiles, LANES, SUB); v3 = vpad.reshape(ntiles, LANES, SUB)
    # ---- head forward (sequential, exact; the kernel does it with a general affine wave scan)
    mu = tab["mu0"].copy()
    r_head = np.zeros(nh)
    ss_head = 0.0
    for t in range(nh):
        r = y[t] - hh - h @ mu
        r_head[t] = r
        ss_head += r * r * tab["iS_t"][t]
        mu = A @ mu + a + tab["kA_t"][t] * r
    # ---- P1: steady tile elements with zero carries
    F = np.zeros((ntiles, d)); B0 = np.zeros((ntiles, d))
    for i in range(th, ntiles):
        F[i], B0[i], _, _ = tile_steady(tab, y3[i], v3[i], np.zeros(d), np.zeros(d), False)
    # ---- S: carries
    M = tab["PhiPow"][9]; Gm = tab["GPow"][9]
    MU = np.zeros((ntiles + 1, d)); MU[th] = mu
    for i in range(th, ntiles):
        MU[i + 1] = M @ MU[i] + F[i]
    LAM = np.zeros((ntiles + 1, d))
    for i in range(ntiles - 1, th - 1, -1):
        LAM[i] = Gm @ LAM[i + 1] + B0[i] - (tab["Bl"] if i == ntiles - 1 else tab["B"]) @ MU[i]
    # ---- P2
    mean = np.zeros(ntiles * TILE)
    ssq = 0.0
    for i in range(th, ntiles):
        _, b, s2, out = tile_steady(tab, y3[i], v3[i], MU[i], LAM[i + 1], True)
        assert np.allclose(b, LAM[i], rtol=1e-9, atol=1e-12), (i, b, LAM[i])
        ssq += s2
        mean[i * TILE:(i + 1) * TILE] = out.reshape(-1)
    var = np.full(ntiles * TILE, tab["vb_ss"])
    # tail table
    jj = np.arange(n1)
    var[T - 1 - jj] = tab["vbt"][:n1]
    # ---- head backward
    lam = LAM[th].copy()
    for t in range(nh - 1, -1, -1):
        mean[t] = y[t] - tab["rS_t"][t] * r_head[t] + h @ lam
        lam = tab["G_t"][t] @ lam + tab["c_t"][t] * r_head[t]
    var[:nh] = tab["vbh"]
    var = var[:T] + Rnew
    lml = -0.5 * (T * LOG2PI + tab["LS_head"] + (T - n0) * tab["ss"]["logS"] + ss_head + tab["ss"]["iS"] * ssq)
    return lml, mean[:T], var, tab


if __name__ == "__main__":
    from oracle import components as oc
    from oracle import lgssm_ref as ref
    from oracle import seq_kalman as sk

    for kern, dt, T in [(("matern52",), 0.1, 3000), (("matern32",), 0.1, 2100), (("matern52", "matern32"), 0.1, 2500),
                        (("matern52",), 0.03, 5000)]:
        model = oc.build_lgssm(kern, ("regular", 0.0, dt, T), 0.1)
        d = np.asarray(model["A"]).shape[-1]
        rng = np.random.default_rng(1)
        y = sk.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
        out = run(model, y, 1e-18)
        if out is None:
            print(kern, dt, "not applicable")
            continue
        lml, mean, var, tab = out
        lp_ref = sk.logpdf(model, y)
        m_ref, v_ref = sk.posterior_marginals(model, y, np.array([1e-18]))
        print(kern, dt, "n0", tab["n0"], "n1", tab["n1"], "lml rel", abs(lml - lp_ref) / abs(lp_ref), "mean", np.max(np.abs(mean - m_ref)),
              "var", np.max(np.abs(var - v_ref)))
