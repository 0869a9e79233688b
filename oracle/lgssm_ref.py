"""ORACLE (test infrastructure only -- never imported by the product path).

Literal NumPy restatement of the reference's linear-Gaussian state-space hot path:

    /root/reference/src/util/scan.jl:15-28                       scan_emit
    /root/reference/src/models/lgssm.jl:65-248                   rand / marginals / logpdf / _filter /
                                                                 posterior / invert_dynamics
    /root/reference/src/models/linear_gaussian_conditionals.jl   predict :46-52, predict_marginals :63-68,
                                                                 conditional_rand :84-87, :241-243,
                                                                 posterior_and_lml :129-141 (Small),
                                                                 :247-257 (Scalar)
    /root/reference/src/models/gauss_markov_model.jl:38-46       eachindex (ordering), getindex
    /root/reference/src/models/missings.jl:8-101                 missing-data transform + compensation
    /root/reference/src/util/gaussian.jl:35-43,61-67             rand(::Gaussian), marginals(::Gaussian)

PARITY UNPINNED (in the sense of the build contract): the reference is Julia, Julia is not in this
image, and the reference's test-suite holds no golden vectors for this path.
What pins this restatement instead are the reference tests' own *identities*, asserted in
tests/test_oracle_identities.py: state-space == dense GP (test/gp/lti_sde.jl:193-200), missing ==
analytically marginalised (test/models/missings.jl:94-115), Scalar == Small with p=1
(test/models/linear_gaussian_conditionals.jl:117-126).

Conventions: a model is a dict
    ordering : 'F' | 'R'                  (Forward / Reverse, gauss_markov_model.jl:1-9)
    A (T|1,d,d)  a (T|1,d)  Q (T|1,d,d)   transitions; leading dim 1 == FillArrays.Fill (shared)
    kind : 'scalar' | 'small' | 'large' | 'bottleneck'   Scalar/Small/LargeOutputLGC emissions (H, h, R) / BottleneckLGC
                                           (Hb, hb = the projection; H, h, R = its fan-out LargeOutputLGC)
    H (T|1,d) [scalar]  or (T|1,p,d)      emission "A" (for scalar: the vector h with A = h')
    h (T|1,)  [scalar]  or (T|1,p)        emission "a"
    R (T|1,)  [scalar]  or (T|1,p,p)      emission "Q"
    x0m (d,), x0P (d,d)                   x0
    T : int
Pure-Python loops: use for small T only (seconds up to T ~ 1e5). oracle/seq_kalman.c is the
large-T restatement.
"""
import numpy as np

LOG2PI = float(np.log(2.0 * np.pi))
LARGE_VAR = 1e15  # missings.jl:43


def _at(arr, t):
    """Fill semantics: a leading dim of 1 is a value shared by all steps."""
    return arr[t] if arr.shape[0] > 1 else arr[0]


def symmetric(P):
    """LinearAlgebra.Symmetric(P): reads the upper triangle only (lgc.jl:50-51)."""
    U = np.triu(P)
    return U + np.triu(P, 1).T


def chol_upper(S):
    """cholesky(Symmetric(S)).U -- upper factor of the upper-triangle-symmetrised matrix."""
    return np.linalg.cholesky(symmetric(S)).T


def idx_order(model):
    """gauss_markov_model.jl:38-40."""
    T = model["T"]
    return range(T) if model["ordering"] == "F" else range(T - 1, -1, -1)


def transition(model, t):
    return _at(model["A"], t), _at(model["a"], t), _at(model["Q"], t)


def emission(model, t):
    return _at(model["H"], t), _at(model["h"], t), _at(model["R"], t)


# ---------------------------------------------------------------------------- per-step maths
def predict(m, P, A, a, Q):
    """lgc.jl:46-52:  Gaussian(A*m + a, (A*symmetric(P))*A' + Q)."""
    return A @ m + a, (A @ symmetric(P)) @ A.T + Q


def predict_emission(model, m, P, t):
    H, h, R = emission(model, t)
    if model["kind"] == "scalar":
        # A = H' is 1xd; result is a scalar Gaussian.
        return float(H @ m + h), float((H @ symmetric(P)) @ H + R)
    if model["kind"] == "bottleneck":                           # lgc.jl:314
        zm, zP = bottleneck_project(m, P, _at(model["Hb"], t), _at(model["hb"], t))
        return predict(zm, zP, H, h, R)
    return predict(m, P, H, h, R)


def posterior_and_lml_scalar(m, P, H, h, R, y):
    """lgc.jl:247-257 (ScalarOutputLGC)."""
    V = H @ P                      # A*P, 1xd (full P, no Symmetric wrapper here)
    sqrtS = np.sqrt(V @ H + R)
    B = V / sqrtS
    alpha = (y - (H @ m + h)) / sqrtS
    lml = -(LOG2PI + 2.0 * np.log(sqrtS) + alpha ** 2) / 2.0
    return m + B * alpha, P - np.outer(B, B), float(lml)


def posterior_and_lml_small(m, P, H, h, R, y):
    """lgc.jl:129-141 (SmallOutputLGC)."""
    V = H @ P
    U = chol_upper(V @ H.T + R)
    B = np.linalg.solve(U.T, V)
    alpha = np.linalg.solve(U.T, y - (H @ m + h))
    logdetS = 2.0 * np.sum(np.log(np.diag(U)))
    lml = -(len(y) * LOG2PI + logdetS + alpha @ alpha) / 2.0
    return m + B.T @ alpha, P - B.T @ B, float(lml)


def posterior_and_lml_large(m, P, A, a, Q, y):
    """lgc.jl:179-204 (LargeOutputLGC): the same conditional as SmallOutputLGC computed through Cholesky factors of Q and
    of P + 1e-10 I (that jitter makes it equal to the Small form only to the reference's own `isapprox` tolerance,
    test/models/linear_gaussian_conditionals.jl:65-75)."""
    d = len(m)
    Qu = chol_upper(symmetric(Q))
    Pu = chol_upper(symmetric(P + 1e-10 * np.eye(d)))
    Bt = np.linalg.solve(Qu.T, A) @ Pu.T                      # Q.U' \ A * P.U'
    Fu = chol_upper(symmetric(Bt.T @ Bt + np.eye(d)))
    G = np.linalg.solve(Fu.T, Pu)
    P_post = G.T @ G
    delta = np.linalg.solve(Qu.T, y - (A @ m + a))
    beta = np.linalg.solve(Fu.T, Bt.T @ delta)
    m_post = m + G.T @ beta
    c = len(y) * LOG2PI
    logdetF = 2.0 * np.sum(np.log(np.diag(Fu)))
    logdetQ = 2.0 * np.sum(np.log(np.diag(Qu)))
    lml = -(delta @ delta - beta @ beta + c + logdetF + logdetQ) / 2.0     # lgc.jl:207
    return m_post, P_post, float(lml)


def bottleneck_project(m, P, Hb, hb):
    """lgc.jl:308-312: Gaussian(H m + h, H P H' + 1e-12 I)."""
    return Hb @ m + hb, Hb @ P @ Hb.T + 1e-12 * np.eye(Hb.shape[0])


def posterior_and_lml_bottleneck(m, P, Hb, hb, A, a, Q, y):
    """lgc.jl:320-336 (BottleneckLGC): posterior over z = Hb x + hb through the fan-out LargeOutputLGC, then
    x | y by integrating x | z against z | y."""
    zm, zP = bottleneck_project(m, P, Hb, hb)
    zpm, zpP, lml = posterior_and_lml_large(zm, zP, A, a, Q, y)
    U = chol_upper(symmetric(zP + 1e-12 * np.eye(len(zm))))
    Gt = np.linalg.solve(U, np.linalg.solve(U.T, Hb @ P))
    return m + Gt.T @ (zpm - zm), P + Gt.T @ (zpP - zP) @ Gt, lml


def small_from_bottleneck(model):
    """test/test_util.jl `small_output_lgc_from_bottleneck`: y | x ~ N(A (Hb x + hb) + a, Q) as one SmallOutputLGC."""
    T = model["T"]
    n = max(model[k].shape[0] for k in ("Hb", "hb", "H", "h"))
    H = np.stack([_at(model["H"], t) @ _at(model["Hb"], t) for t in range(n)])
    h = np.stack([_at(model["H"], t) @ _at(model["hb"], t) + _at(model["h"], t) for t in range(n)])
    out = {k: v for k, v in model.items() if k not in ("Hb", "hb")}
    out.update(kind="small", H=H, h=h)
    return out


def posterior_and_lml(model, m, P, t, y):
    H, h, R = emission(model, t)
    if model["kind"] == "scalar":
        return posterior_and_lml_scalar(m, P, H, h, R, y)
    if model["kind"] == "large":
        return posterior_and_lml_large(m, P, H, h, R, y)
    if model["kind"] == "bottleneck":
        return posterior_and_lml_bottleneck(m, P, _at(model["Hb"], t), _at(model["hb"], t), H, h, R, y)
    return posterior_and_lml_small(m, P, H, h, R, y)


def invert_dynamics(mf, Pf, mp, Pp, A):
    """lgssm.jl:231-238. Returns (G, g, L) of the time-reversed transition."""
    d = len(mf)
    U = chol_upper(Pp + 1e-10 * np.eye(d))
    Gt = np.linalg.solve(U, np.linalg.solve(U.T, A @ Pf))
    UG = U @ Gt
    return Gt.T.copy(), mf - Gt.T @ mp, Pf - UG.T @ UG


# ---------------------------------------------------------------------------- T-step algorithms
def logpdf_terms(model, ys):
    """lgssm.jl:147-165: per-step lml (emitted) and the final state."""
    m, P = model["x0m"].copy(), model["x0P"].copy()
    out = np.zeros(model["T"])
    fwd = model["ordering"] == "F"
    for t in idx_order(model):
        A, a, Q = transition(model, t)
        if fwd:
            m, P = predict(m, P, A, a, Q)
            m, P, lml = posterior_and_lml(model, m, P, t, ys[t])
        else:
            m, P, lml = posterior_and_lml(model, m, P, t, ys[t])
            m, P = predict(m, P, A, a, Q)
        out[t] = lml
    return out, (m, P)


def logpdf(model, ys):
    """lgssm.jl:147-151: plain left-to-right sum in index order."""
    terms, _ = logpdf_terms(model, ys)
    acc = 0.0
    for t in range(model["T"]):  # `sum` over the emitted vector (storage order)
        acc += terms[t]
    return acc


def filter_(model, ys):
    """lgssm.jl:171-187 (_filter): filtering distributions (ms (T,d), Ps (T,d,d))."""
    T, d = model["T"], len(model["x0m"])
    ms, Ps = np.zeros((T, d)), np.zeros((T, d, d))
    m, P = model["x0m"].copy(), model["x0P"].copy()
    fwd = model["ordering"] == "F"
    for t in idx_order(model):
        A, a, Q = transition(model, t)
        if fwd:
            m, P = predict(m, P, A, a, Q)
            m, P, _ = posterior_and_lml(model, m, P, t, ys[t])
            ms[t], Ps[t] = m, P
        else:
            m, P, _ = posterior_and_lml(model, m, P, t, ys[t])
            ms[t], Ps[t] = m, P
            m, P = predict(m, P, A, a, Q)
    return ms, Ps


def posterior(model, ys):
    """lgssm.jl:193-228: LGSSM of the opposite ordering with transitions (G, g, L), x0 = final state."""
    if model["T"] != len(ys):
        raise ValueError(
            f"Dimension mismatch. length(prior) is {model['T']}, but length(y) is {len(ys)}")
    T, d = model["T"], len(model["x0m"])
    G, g, L = np.zeros((T, d, d)), np.zeros((T, d)), np.zeros((T, d, d))
    m, P = model["x0m"].copy(), model["x0P"].copy()
    fwd = model["ordering"] == "F"
    for t in idx_order(model):
        A, a, Q = transition(model, t)
        if fwd:
            mp, Pp = predict(m, P, A, a, Q)
            G[t], g[t], L[t] = invert_dynamics(m, P, mp, Pp, A)
            m, P, _ = posterior_and_lml(model, mp, Pp, t, ys[t])
        else:
            mf, Pf, _ = posterior_and_lml(model, m, P, t, ys[t])
            mp, Pp = predict(mf, Pf, A, a, Q)
            # invert_dynamics(xp, xf, t): roles swapped exactly as lgssm.jl:227
            G[t], g[t], L[t] = invert_dynamics(mp, Pp, mf, Pf, A)
            m, P = mp, Pp
    post = dict(model)
    post.update(ordering="R" if fwd else "F", A=G, a=g, Q=L, x0m=m, x0P=P)
    return post


def marginals(model):
    """lgssm.jl:99-115: emission marginals (mean, cov) at every step.
    scalar kind -> (T,), (T,);  small kind -> (T,p), (T,p,p)."""
    T = model["T"]
    m, P = model["x0m"].copy(), model["x0P"].copy()
    fwd = model["ordering"] == "F"
    means, covs = [None] * T, [None] * T
    for t in idx_order(model):
        A, a, Q = transition(model, t)
        if fwd:
            m, P = predict(m, P, A, a, Q)
            means[t], covs[t] = predict_emission(model, m, P, t)
        else:
            means[t], covs[t] = predict_emission(model, m, P, t)
            m, P = predict(m, P, A, a, Q)
    return np.array(means), np.array(covs)


def latent_marginals(model):
    """Same recursion as `marginals`, but returns the latent (m_t, P_t) the emission predict sees."""
    T, d = model["T"], len(model["x0m"])
    ms, Ps = np.zeros((T, d)), np.zeros((T, d, d))
    m, P = model["x0m"].copy(), model["x0P"].copy()
    fwd = model["ordering"] == "F"
    for t in idx_order(model):
        A, a, Q = transition(model, t)
        if fwd:
            m, P = predict(m, P, A, a, Q)
            ms[t], Ps[t] = m, P
        else:
            ms[t], Ps[t] = m, P
            m, P = predict(m, P, A, a, Q)
    return ms, Ps


def conditional_rand_transition(eps, A, a, Q, x):
    """lgc.jl:84-87: (A*x + a) + cholesky(symmetric(Q + 1e-9 I)).U' * eps."""
    d = len(a)
    U = chol_upper(Q + 1e-9 * np.eye(d))
    return (A @ x + a) + U.T @ eps


def conditional_rand_emission(model, eps, t, x):
    H, h, R = emission(model, t)
    if model["kind"] == "scalar":
        return float((H @ x + h) + np.sqrt(R) * eps)          # lgc.jl:241-243
    if model["kind"] == "bottleneck":                            # lgc.jl:297-300
        x = _at(model["Hb"], t) @ x + _at(model["hb"], t)
    return conditional_rand_transition(eps, H, h, R, x)         # lgc.jl:84-87 (p-dim)


def rand_x0(eps0, m, P):
    """gaussian.jl:35-43: mean + cholesky(Symmetric(P + 1e-12 I)).U' * eps."""
    U = chol_upper(P + 1e-12 * np.eye(len(m)))
    return m + U.T @ eps0


def rand(model, eps_t, eps_e, eps_0):
    """lgssm.jl:65-91 with the randomness supplied (eps_t (T,d), eps_e (T,)|(T,p), eps_0 (d,)).
    Noise for step t is indexed by t (storage order), exactly as zip(eps, model) does."""
    T = model["T"]
    x = rand_x0(eps_0, model["x0m"], model["x0P"])
    fwd = model["ordering"] == "F"
    ys = [None] * T
    for t in idx_order(model):
        A, a, Q = transition(model, t)
        if fwd:
            x = conditional_rand_transition(eps_t[t], A, a, Q, x)
            ys[t] = conditional_rand_emission(model, eps_e[t], t, x)
        else:
            ys[t] = conditional_rand_emission(model, eps_e[t], t, x)
            x = conditional_rand_transition(eps_t[t], A, a, Q, x)
    return np.array(ys)


# ---------------------------------------------------------------------------- missing data
def replace_observation_noise_cov(model, R_new):
    """missings.jl:35-41."""
    out = dict(model)
    out["R"] = np.asarray(R_new, dtype=np.float64)
    return out


def _densify_R(model):
    T = model["T"]
    R = model["R"]
    if R.shape[0] == 1 and T > 1:
        R = np.repeat(R, T, axis=0)      # missings.jl:83-85 collect(Fill)
    return R.copy()


def transform_model_and_obs(model, ys, missing):
    """missings.jl:25-33,55-101. `missing` is a bool mask (T,) for whole-step missing (scalar and
    small kinds) or (T,p) for per-element missing (small kind with diagonal R only, lgc.jl:143-151)."""
    missing = np.asarray(missing, dtype=bool)
    R = _densify_R(model)
    ys = np.array(ys, dtype=np.float64, copy=True)
    if missing.ndim == 1:
        for t in np.nonzero(missing)[0]:
            if model["kind"] == "scalar":
                R[t] = LARGE_VAR
                ys[t] = 0.0
            else:
                p = R.shape[-1]
                R[t] = LARGE_VAR * np.eye(p)
                ys[t] = 0.0
        n_missing = int(missing.sum()) * (1 if model["kind"] == "scalar" else R.shape[-1])
    else:
        if model["kind"] == "scalar":
            raise TypeError("per-element missing needs vector observations")
        for t, j in zip(*np.nonzero(missing)):
            off = R[t] - np.diag(np.diag(R[t]))
            if np.any(off != 0.0):
                raise TypeError("MethodError: per-element missing requires Diagonal noise (lgc.jl:146)")
            R[t, j, j] = LARGE_VAR
            ys[t, j] = 0.0
        n_missing = int(missing.sum())
    return replace_observation_noise_cov(model, R), ys, n_missing


def volume_compensation(n_missing):
    """missings.jl:45-53."""
    return n_missing * np.log(2.0 * np.pi * LARGE_VAR) / 2.0


def logpdf_missing(model, ys, missing):
    """missings.jl:8-13."""
    m2, y2, n = transform_model_and_obs(model, ys, missing)
    return logpdf(m2, y2) + volume_compensation(n)


def filter_missing(model, ys, missing):
    m2, y2, _ = transform_model_and_obs(model, ys, missing)
    return filter_(m2, y2)


def posterior_missing(model, ys, missing):
    m2, y2, _ = transform_model_and_obs(model, ys, missing)
    return posterior(m2, y2)


# ---------------------------------------------------------------------------- modified Bryson-Frazier form (checker of the
# persistent mid-d backward pass, temporalgps.jl_amd/csrc/tgp_dense_fused.hpp; NOT a restatement of reference code)
def bryson_frazier_marginals(model, ys, R_new, missing=None):
    """Emission marginals of the smoothed states of a Forward scalar/diagonal-noise model by the adjoint recursion
        per scalar update (reverse):  K = v / s,  Lam <- (I - K h)' Lam (I - K h) + h'h / s,  lam <- (I - K h)' lam - h' nu / s
        per transition:               Lam <- A' Lam A,  lam <- A' lam
        mean = h m_f + hh - w'lam,  var = h w - w' Lam w + R_new,   w = P_f h'
    -- the exact posterior of the model; the reference's reverse-time form (lgssm.jl:193-238) differs from it by the effect of its
    1e-10 jitter. Observations are applied one scalar at a time (diagonal noise), missing ones as y := 0, R := 1e15."""
    T, d = model["T"], len(model["x0m"])
    assert model["ordering"] == "F"
    m, P = model["x0m"].copy(), model["x0P"].copy()
    rec = []
    for t in range(T):
        A, a, Q = transition(model, t)
        H, h, R = emission(model, t)
        H, h = np.atleast_2d(H), np.atleast_1d(h)
        Rd = np.atleast_1d(R) if np.ndim(R) < 2 else np.diagonal(R)
        yt = np.atleast_1d(ys[t])
        m = A @ m + a
        P = A @ P @ A.T + Q
        upd = []
        for j in range(len(h)):
            miss = missing is not None and bool(np.atleast_1d(missing[t])[j if np.ndim(missing[t]) else 0])
            v = P @ H[j]
            s = H[j] @ v + (1e15 if miss else Rd[j])
            nu = (0.0 if miss else yt[j]) - H[j] @ m - h[j]
            m = m + v * nu / s
            P = P - np.outer(v, v) / s
            upd.append((v, s, nu))
        rec.append((m.copy(), P.copy(), upd, A, H, h))
    lam, Lam = np.zeros(d), np.zeros((d, d))
    p = len(rec[0][5])
    mean, var = np.zeros((T, p)), np.zeros((T, p))
    Rn = np.broadcast_to(np.asarray(R_new, dtype=np.float64).reshape(-1, p) if np.ndim(R_new) else np.full((1, p), float(R_new)), (T, p))
    for t in range(T - 1, -1, -1):
        mf, Pf, upd, A, H, h = rec[t]
        for j in range(p):
            w = Pf @ H[j]
            mean[t, j] = H[j] @ mf + h[j] - w @ lam
            var[t, j] = H[j] @ w - w @ Lam @ w + Rn[t, j]
        for j in range(p - 1, -1, -1):
            v, s, nu = upd[j]
            C = np.eye(d) - np.outer(v / s, H[j])
            Lam = C.T @ Lam @ C + np.outer(H[j], H[j]) / s
            lam = C.T @ lam - H[j] * nu / s
        Lam = A.T @ Lam @ A
        lam = A.T @ lam
    return (mean[:, 0], var[:, 0]) if p == 1 else (mean, var)
