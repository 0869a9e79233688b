"""ORACLE HARDENING (test infrastructure): the scalar-output recursions of oracle/lgssm_ref.py -- predict (lgc.jl:46-52),
posterior_and_lml(ScalarOutputLGC) (lgc.jl:247-257), invert_dynamics with its 1e-10 jitter (lgssm.jl:231-238) and the Reverse
step_marginals (lgssm.jl:111-115) -- evaluated in 50-digit arithmetic (mpmath). Same algorithm, same operation order, no
fp64 rounding: the difference between this and oracle/lgssm_ref.py IS the fp64 oracle's own rounding error, which bounds how
tight a parity tolerance against the oracle can meaningfully be. It does not pin the oracle to the reference (PARITY UNPINNED
stays: the Julia package cannot run here); it removes "the oracle's rounding" from the list of unknowns."""
import mpmath as mp
import numpy as np

mp.mp.dps = 50


def _M(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        return mp.matrix([mp.mpf(float(x)) for x in a])
    return mp.matrix([[mp.mpf(float(x)) for x in row] for row in a])


def _at(arr, t):
    arr = np.asarray(arr)
    return arr[t] if arr.shape[0] > 1 else arr[0]


def _sym_upper(P):
    """Symmetric(P): the upper triangle mirrored (lgc.jl:50)."""
    n = P.rows
    S = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            S[i, j] = P[min(i, j), max(i, j)]
    return S


def predict(m, P, A, a, Q):
    return A * m + a, (A * _sym_upper(P)) * A.T + Q


def update_scalar(m, P, H, h, R, y):
    V = (H.T * P)                       # 1 x d
    s2 = (V * H)[0] + R
    sq = mp.sqrt(s2)
    B = V / sq
    alpha = (y - ((H.T * m)[0] + h)) / sq
    lml = -(mp.log(2 * mp.pi) + 2 * mp.log(sq) + alpha ** 2) / 2
    return m + B.T * alpha, P - B.T * B, lml


def _tri_solve_lower(L, Bm):
    n, k = L.rows, Bm.cols
    X = mp.matrix(n, k)
    for c in range(k):
        for i in range(n):
            s = Bm[i, c]
            for j in range(i):
                s -= L[i, j] * X[j, c]
            X[i, c] = s / L[i, i]
    return X


def _tri_solve_upper(U, Bm):
    n, k = U.rows, Bm.cols
    X = mp.matrix(n, k)
    for c in range(k):
        for i in range(n - 1, -1, -1):
            s = Bm[i, c]
            for j in range(i + 1, n):
                s -= U[i, j] * X[j, c]
            X[i, c] = s / U[i, i]
    return X


def invert_dynamics(mf, Pf, mp_, Pp, A):
    n = Pp.rows
    J = _sym_upper(Pp + mp.mpf("1e-10") * mp.eye(n))
    Lc = mp.cholesky(J)                 # lower; U = Lc'
    U = Lc.T
    Gt = _tri_solve_upper(U, _tri_solve_lower(Lc, A * Pf))
    G = Gt.T
    UG = U * Gt
    return G, mf - G * mp_, Pf - UG.T * UG


def run(model, y, R_new=None, missing=None):
    """Forward-ordered scalar model. Returns dict(logpdf, post_mean, post_var) as Python floats / arrays (rounded ONCE)."""
    T = int(model["T"])
    m, P = _M(model["x0m"]), _M(model["x0P"])
    lml = mp.mpf(0)
    rev = []
    nmiss = 0
    for t in range(T):
        A, a, Q = _M(_at(model["A"], t)), _M(_at(model["a"], t)), _M(_at(model["Q"], t))
        H, h, R = _M(_at(model["H"], t)), mp.mpf(float(_at(np.atleast_1d(model["h"]), t))), mp.mpf(float(_at(np.atleast_1d(model["R"]), t)))
        yt = mp.mpf(float(y[t]))
        if missing is not None and missing[t]:
            yt, R = mp.mpf(0), mp.mpf("1e15")
            nmiss += 1
        mpred, Ppred = predict(m, P, A, a, Q)
        if R_new is not None:
            rev.append(invert_dynamics(m, P, mpred, Ppred, A))
        m, P, l = update_scalar(mpred, Ppred, H, h, R, yt)
        lml += l
    out = dict(logpdf=float(lml + nmiss * mp.log(2 * mp.pi * mp.mpf("1e15")) / 2))
    if R_new is not None:
        Rn = np.atleast_1d(np.asarray(R_new, dtype=np.float64))
        mean, var = np.zeros(T), np.zeros(T)
        x, Px = m, P
        for t in range(T - 1, -1, -1):
            H, h = _M(_at(model["H"], t)), mp.mpf(float(_at(np.atleast_1d(model["h"]), t)))
            rn = mp.mpf(float(Rn[t] if Rn.shape[0] > 1 else Rn[0]))
            mean[t] = float((H.T * x)[0] + h)
            var[t] = float(((H.T * _sym_upper(Px)) * H)[0] + rn)
            G, g, Lq = rev[t]
            x, Px = G * x + g, (G * _sym_upper(Px)) * G.T + Lq
        out.update(post_mean=mean, post_var=var)
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# 50-digit logpdf of a sum of scaled, stretched Matern kernels on a regular grid, INCLUDING the map hyper-parameters -> blocks
# (lti_sde.jl:148-160: A = exp(F s dt), Q = P - A P A'; :205-235 the Matern tables; :324-346 scaling; :350-373 stretching; :404-422 sums),
# and its gradient by central differences in the same arithmetic (step 1e-20 relative: truncation ~1e-40, rounding ~1e-30) -- the
# reference value the device gradients are held against (tests/test_gpu_gradient.py). Test infrastructure.
def _matern_mp(name):
    if name == "matern12":
        return mp.matrix([[-1]]), mp.matrix([[1]])
    if name == "matern32":
        lam = mp.sqrt(3)
        return mp.matrix([[0, 1], [-3, -2 * lam]]), mp.matrix([[1, 0], [0, 3]])
    if name == "matern52":
        lam, kap = mp.sqrt(5), mp.mpf(5) / 3
        return mp.matrix([[0, 1, 0], [0, 0, 1], [-lam ** 3, -3 * lam ** 2, -3 * lam]]), mp.matrix([[1, 0, -kap], [0, kap, 0], [-kap, 0, 25]])
    raise ValueError(name)


def lti_blocks_mp(terms, dt):
    """terms: [(name, sigma2, stretch), ...] -> (A, Q, H, P0) as mp matrices (block diagonal / concatenated)"""
    parts = []
    for name, s2, s in terms:
        F, P = _matern_mp(name)
        A = mp.expm(F * (mp.mpf(s) * mp.mpf(dt)))
        parts.append((A, P - A * P * A.T, mp.sqrt(mp.mpf(s2)), P))
    d = sum(p[0].rows for p in parts)
    A, Q, P0, H = mp.zeros(d, d), mp.zeros(d, d), mp.zeros(d, d), mp.zeros(d, 1)
    o = 0
    for Ai, Qi, sig, Pi in parts:
        n = Ai.rows
        for i in range(n):
            for j in range(n):
                A[o + i, o + j], Q[o + i, o + j], P0[o + i, o + j] = Ai[i, j], Qi[i, j], Pi[i, j]
        H[o, 0] = sig
        o += n
    return A, Q, H, P0


def lti_logpdf_mp(terms, dt, sigma2, y):
    A, Q, H, P = lti_blocks_mp(terms, dt)
    d = A.rows
    m, a, R, h = mp.zeros(d, 1), mp.zeros(d, 1), mp.mpf(sigma2), mp.mpf(0)
    lml = mp.mpf(0)
    for yt in y:
        mpred, Ppred = predict(m, P, A, a, Q)
        m, P, l = update_scalar(mpred, Ppred, H, h, R, mp.mpf(float(yt)))
        lml += l
    return lml


def lti_gradient_mp(terms, dt, sigma2, y, rel=mp.mpf("1e-20")):
    """-> (logpdf, [d / d sigma2_i, d / d stretch_i for every term ..., d / d noise]) as Python floats"""
    def f(vec):
        tt = [(terms[i][0], vec[2 * i], vec[2 * i + 1]) for i in range(len(terms))]
        return lti_logpdf_mp(tt, dt, vec[-1], y)
    v0 = [mp.mpf(x) for t in terms for x in (t[1], t[2])] + [mp.mpf(sigma2)]
    grad = []
    for k in range(len(v0)):
        hs = rel * v0[k]
        vp, vm = list(v0), list(v0)
        vp[k] += hs
        vm[k] -= hs
        grad.append(float((f(vp) - f(vm)) / (2 * hs)))
    return float(f(v0)), grad
