"""ORACLE (test infrastructure only -- never imported by the product path).

NumPy/SciPy restatement of the host-side component construction that FEEDS the hot path:

    /root/reference/src/gp/lti_sde.jl:71-80      build_lgssm
    /root/reference/src/gp/lti_sde.jl:88-109     build_emissions (real hs => ScalarOutputLGC)
    /root/reference/src/gp/lti_sde.jl:112-131    mean functions (h_t += m(t))
    /root/reference/src/gp/lti_sde.jl:135-160    broadcast_components (irregular / regular)
    /root/reference/src/gp/lti_sde.jl:189-235    Matern12/32/52 SDEs and stationary distributions
    /root/reference/src/gp/lti_sde.jl:239-252    Cosine
    /root/reference/src/gp/lti_sde.jl:255-307    ApproxPeriodicKernel
    /root/reference/src/gp/lti_sde.jl:311-321    Constant
    /root/reference/src/gp/lti_sde.jl:324-346    Scaled   (scales H,h by sigma, not P_inf)
    /root/reference/src/gp/lti_sde.jl:350-373    Stretched (ScaleTransform: stretches the inputs)
    /root/reference/src/gp/lti_sde.jl:377-400    Product (Kronecker sum)
    /root/reference/src/gp/lti_sde.jl:404-445    Sum (block diagonal)
    /root/reference/src/gp/posterior_lti_sde.jl:18-158  posterior queries (merge/sort/missing)

PARITY UNPINNED vs reference-run outputs (no Julia here, no golden vectors in the reference); pinned by
the reference's state-space == dense-GP identities (tests/test_oracle_identities.py, oracle/dense_gp.py).
Third-party arithmetic: matrix exponential = scipy.linalg.expm (the reference uses StaticArrays /
LinearAlgebra `exp`, call sites lti_sde.jl:140,152).

Kernel specs are nested tuples:
    ("matern12",) ("matern32",) ("matern52",) ("cosine",) ("constant", c)
    ("approx_periodic", N, r) ("scaled", sigma2, k) ("stretched", s, k) ("sum", k1, k2, ...)
    ("product", k1, k2, ...)
Inputs: `t` is either a 1-D array (irregular => per-step blocks) or ("regular", t0, dt, N)
(RegularSpacing / StepRangeLen => Fill blocks, leading dim 1).
"""
import numpy as np
from scipy.linalg import expm, block_diag
from scipy.special import ive

from . import lgssm_ref as ref


# ------------------------------------------------------------------ inputs
def is_regular(t):
    return isinstance(t, tuple) and t[0] == "regular"


def times(t):
    if is_regular(t):
        _, t0, dt, N = t
        return t0 + np.arange(N) * dt          # regular_data.jl:18
    return np.asarray(t, dtype=np.float64)


def n_times(t):
    return t[3] if is_regular(t) else len(t)


def apply_stretch(s, t):
    if is_regular(t):
        _, t0, dt, N = t
        return ("regular", s * t0, s * dt, N)  # lti_sde.jl:373
    return s * np.asarray(t, dtype=np.float64)


# ------------------------------------------------------------------ base SDEs
def _colmajor(n, vals):
    return np.array(vals, dtype=np.float64).reshape(n, n, order="F")


def to_sde(k):
    """(F, q, H) -- lti_sde.jl:189-252, :311-346, :350-353."""
    name = k[0]
    if name == "matern12":
        return np.array([[-1.0]]), 2.0, np.array([1.0])
    if name == "matern32":
        lam = np.sqrt(3.0)
        return _colmajor(2, [0, -3, 1, -2 * lam]), 4 * lam ** 3, np.array([1.0, 0.0])
    if name == "matern52":
        lam = np.sqrt(5.0)
        F = _colmajor(3, [0, 0, -lam ** 3, 1, 0, -3 * lam ** 2, 0, 1, -3 * lam])
        return F, 8 * lam ** 5 / 3, np.array([1.0, 0.0, 0.0])
    if name == "cosine":
        return _colmajor(2, [0, 1, -1, 0]), 0.0, np.array([1.0, 0.0])
    if name == "constant":
        return np.array([[0.0]]), 0.0, np.array([1.0])
    if name == "approx_periodic":
        N = k[1]
        F0, _, H0 = to_sde(("cosine",))
        F = block_diag(*[2 * np.pi * i * F0 for i in range(N)])
        return F, 0.0, np.tile(H0, N)
    if name == "scaled":
        F, q, H = to_sde(k[2])
        sig = np.sqrt(k[1])
        return F, sig ** 2 * q, sig * H
    if name == "stretched":
        F, q, H = to_sde(k[2])
        return F * k[1], q, H
    raise ValueError(f"to_sde: unsupported kernel {name}")


def stationary_distribution(k):
    """(m, P) -- lti_sde.jl:199-203,213-218,230-235,246-250,291-307,318-321,330-332,355-359."""
    name = k[0]
    if name == "matern12":
        return np.zeros(1), np.array([[1.0]])
    if name == "matern32":
        return np.zeros(2), np.diag([1.0, 3.0])
    if name == "matern52":
        kap = 5.0 / 3.0
        return np.zeros(3), _colmajor(3, [1, 0, -kap, 0, kap, 0, -kap, 0, 25])
    if name == "cosine":
        return np.zeros(2), np.eye(2)
    if name == "constant":
        return np.zeros(1), np.array([[float(k[1])]])
    if name == "approx_periodic":
        N, r = k[1], k[2]
        l2 = 1.0 / (4.0 * r ** 2)
        # besseli(j-1, l2) / exp(l2) == ive(j-1, l2) for l2 > 0
        Ps = [(1 + (j != 1)) * ive(j - 1, l2) * np.eye(2) for j in range(1, N + 1)]
        return np.zeros(2 * N), block_diag(*Ps)
    if name in ("scaled", "stretched"):
        return stationary_distribution(k[2])
    raise ValueError(f"stationary_distribution: unsupported kernel {name}")


# ------------------------------------------------------------------ discretisation
def broadcast_components(FqH, x0, t):
    """lti_sde.jl:135-160. Returns A, a, Q, H, h with leading dim T (irregular) or 1 (regular)."""
    F, _, H = FqH
    m0, P0 = x0
    P = ref.symmetric(P0)
    d = F.shape[0]
    if is_regular(t):
        _, _, dt, _ = t
        A = expm(F * dt)
        Q = P - A @ P @ A.T
        return A[None], np.zeros((1, d)), Q[None], H[None].copy(), np.zeros(1)
    tt = np.asarray(t, dtype=np.float64)
    tt = np.concatenate([[tt[0] - 1.0], tt])          # lti_sde.jl:139: first dt == 1
    dts = np.diff(tt)
    A = np.stack([expm(F * dt) for dt in dts])
    Q = np.stack([P - Ai @ P @ Ai.T for Ai in A])
    return A, np.zeros((1, d)), Q, H[None].copy(), np.zeros(1)


def _expand(arr, T):
    return arr if arr.shape[0] == T else np.repeat(arr, T, axis=0)


def lgssm_components(k, t):
    """(A, a, Q, H, h, (m0, P0)); kernel algebra per lti_sde.jl:334-445."""
    name = k[0]
    T = n_times(t)
    if name == "scaled":
        A, a, Q, H, h, x0 = lgssm_components(k[2], t)
        sig = np.sqrt(k[1])
        return A, a, Q, sig * H, sig * h, x0
    if name == "stretched":
        return lgssm_components(k[2], apply_stretch(k[1], t))
    if name == "sum":
        parts = [lgssm_components(kk, t) for kk in k[1:]]
        shared = all(p[0].shape[0] == 1 for p in parts)
        n = 1 if shared else T
        A = np.stack([block_diag(*[_expand(p[0], n)[i] for p in parts]) for i in range(n)])
        Q = np.stack([block_diag(*[_expand(p[2], n)[i] for p in parts]) for i in range(n)])
        na = max(p[1].shape[0] for p in parts)
        a = np.concatenate([_expand(p[1], na) for p in parts], axis=1)
        nh = max(p[3].shape[0] for p in parts)
        H = np.concatenate([_expand(p[3], nh) for p in parts], axis=1)
        nhh = max(p[4].shape[0] for p in parts)
        h = sum(_expand(p[4], nhh) for p in parts)
        m0 = np.concatenate([p[5][0] for p in parts])
        P0 = block_diag(*[p[5][1] for p in parts])
        return A, a, Q, H, h, (m0, P0)
    if name == "product":
        sdes = [to_sde(kk) for kk in k[1:]]
        F = sdes[0][0]
        for s in sdes[1:]:
            B = s[0]
            F = np.kron(F, np.eye(B.shape[0])) + np.kron(np.eye(F.shape[0]), B)   # _kron_add
        q = float(np.prod([s[1] for s in sdes]))
        H = sdes[0][2]
        for s in sdes[1:]:
            H = np.kron(H, s[2])
        x0s = [stationary_distribution(kk) for kk in k[1:]]
        m0, P0 = x0s[0]
        for (mm, PP) in x0s[1:]:
            m0 = np.kron(m0, mm)
            P0 = np.kron(P0, PP)
        A, a, Q, Hs, hs = broadcast_components((F, q, H), (m0, P0), t)
        return A, a, Q, Hs, hs, (m0, P0)
    # SimpleKernel
    x0 = stationary_distribution(k)
    A, a, Q, H, h = broadcast_components(to_sde(k), x0, t)
    return A, a, Q, H, h, x0


def mean_vector(mean, t):
    """AbstractGPs mean functions: None/("zero",), ("const", c), ("custom", f)."""
    tt = times(t)
    if mean is None or mean[0] == "zero":
        return None
    if mean[0] == "const":
        return np.full(len(tt), float(mean[1]))
    if mean[0] == "custom":
        return np.array([mean[1](x) for x in tt], dtype=np.float64)
    raise ValueError(mean)


def build_lgssm(k, t, sigma2, mean=None):
    """lti_sde.jl:71-80 for scalar-output GPs. `sigma2`: scalar (Fill) or (T,) array."""
    A, a, Q, H, h, (m0, P0) = lgssm_components(k, t)
    T = n_times(t)
    mv = mean_vector(mean, t)
    if mv is not None:
        h = _expand(h, T) + mv                           # lti_sde.jl:126-127
    R = np.atleast_1d(np.asarray(sigma2, dtype=np.float64))
    return dict(ordering="F", kind="scalar", T=T, A=A, a=a, Q=Q, H=H, h=h, R=R,
                x0m=np.asarray(m0, dtype=np.float64), x0P=np.asarray(P0, dtype=np.float64))


# ------------------------------------------------------------------ GP-level API (lti_sde.jl:33-68)
def gp_logpdf(k, t, sigma2, y, mean=None, missing=None):
    model = build_lgssm(k, t, sigma2, mean)
    if missing is not None and np.any(missing):
        return ref.logpdf_missing(model, y, missing)
    return ref.logpdf(model, y)


def gp_marginals(k, t, sigma2, mean=None):
    return ref.marginals(build_lgssm(k, t, sigma2, mean))


def gp_rand(k, t, sigma2, eps_t, eps_e, eps_0, mean=None):
    return ref.rand(build_lgssm(k, t, sigma2, mean), eps_t, eps_e, eps_0)


# ------------------------------------------------------------------ posterior (posterior_lti_sde.jl)
def merge_datasets(x1, x2, s1, s2, y1, y2, miss1, miss2):
    """posterior_lti_sde.jl:97-123 (missing carried as a mask alongside y)."""
    x_raw = np.concatenate([x1, x2])
    sort_idx = np.argsort(x_raw, kind="stable")
    x = x_raw[sort_idx]
    S = np.concatenate([s1, s2])[sort_idx]
    y = np.concatenate([y1, y2])[sort_idx]
    miss = np.concatenate([miss1, miss2])[sort_idx]
    inv = np.argsort(sort_idx, kind="stable")
    return x, S, y, miss, inv[: len(y1)], inv[len(y1):]


def _noise_vec(sigma2, n):
    s = np.atleast_1d(np.asarray(sigma2, dtype=np.float64))
    return s if len(s) == n else np.full(n, s[0])


def posterior_marginals(k, x_tr, sigma2_tr, y_tr, x_pr=None, sigma2_pr=1e-18, mean=None,
                        missing_tr=None):
    """marginals(posterior(fx, y)(x_pr, sigma2_pr)) -- posterior_lti_sde.jl:18-37.
    x_pr None or identical to x_tr => same-inputs branch (:27-36). Returns (mean, var)."""
    ntr = n_times(x_tr)
    miss_tr = np.zeros(ntr, dtype=bool) if missing_tr is None else np.asarray(missing_tr, bool)
    same = x_pr is None or (
        n_times(x_pr) == ntr and np.array_equal(times(x_pr), times(x_tr)))
    if same:
        model = build_lgssm(k, x_tr, sigma2_tr, mean)
        post = ref.posterior_missing(model, y_tr, miss_tr) if miss_tr.any() else ref.posterior(model, y_tr)
        post = ref.replace_observation_noise_cov(post, _noise_vec(sigma2_pr, ntr))
        return ref.marginals(post)
    xt, xp = times(x_tr), times(x_pr)
    npr = len(xp)
    x, S, y, miss, tr_idx, pr_idx = merge_datasets(
        xt, xp, _noise_vec(sigma2_tr, ntr), np.full(npr, ref.LARGE_VAR),
        np.asarray(y_tr, dtype=np.float64), np.zeros(npr), miss_tr, np.ones(npr, dtype=bool))
    model = build_lgssm(k, x, S, mean)
    post = ref.posterior_missing(model, y, miss)
    s_full = np.zeros(len(x))
    s_full[pr_idx] = _noise_vec(sigma2_pr, npr)         # build_prediction_obs_vars :136-144
    post = ref.replace_observation_noise_cov(post, s_full)
    mu, var = ref.marginals(post)
    return mu[pr_idx], var[pr_idx]


def posterior_rand(k, x_tr, sigma2_tr, y_tr, x_pr, sigma2_pr, eps_t, eps_e, eps_0, mean=None):
    """rand(rng, posterior(fx,y)(x_pr, sigma2_pr)) with supplied noise -- posterior_lti_sde.jl:48-58."""
    xt, xp = times(x_tr), times(x_pr)
    ntr, npr = len(xt), len(xp)
    x, S, y, miss, tr_idx, pr_idx = merge_datasets(
        xt, xp, _noise_vec(sigma2_tr, ntr), np.full(npr, ref.LARGE_VAR),
        np.asarray(y_tr, dtype=np.float64), np.zeros(npr), np.zeros(ntr, bool), np.ones(npr, bool))
    model = build_lgssm(k, x, S, mean)
    post = ref.posterior_missing(model, y, miss)
    s_full = np.zeros(len(x))
    s_full[pr_idx] = _noise_vec(sigma2_pr, npr)
    post = ref.replace_observation_noise_cov(post, s_full)
    return ref.rand(post, eps_t, eps_e, eps_0)[pr_idx]


def posterior_logpdf(k, x_tr, sigma2_tr, y_tr, x_pr, sigma2_pr, y_pr, mean=None):
    """logpdf(posterior(fx,y)(x_pr, sigma2_pr), y_pr) -- posterior_lti_sde.jl:62-78."""
    xt, xp = times(x_tr), times(x_pr)
    ntr, npr = len(xt), len(xp)
    s_pr = _noise_vec(sigma2_pr, npr)
    x, S, y, miss, tr_idx, pr_idx = merge_datasets(
        xt, xp, _noise_vec(sigma2_tr, ntr), s_pr,
        np.asarray(y_tr, dtype=np.float64), np.zeros(npr), np.zeros(ntr, bool), np.ones(npr, bool))
    model = build_lgssm(k, x, S, mean)
    post = ref.posterior_missing(model, y, miss)
    s_full = np.zeros(len(x))
    s_full[pr_idx] = s_pr
    post = ref.replace_observation_noise_cov(post, s_full)
    y_full = np.zeros(len(x))
    y_full[pr_idx] = y_pr                                # build_prediction_obs :148-158
    miss_full = np.zeros(len(x), dtype=bool)
    miss_full[tr_idx] = True
    return ref.logpdf_missing(post, y_full, miss_full)


# ------------------------------------------------------------------ separable space-time (space_time/to_gauss_markov.jl)
def build_lgssm_separable(k_space, k_time, r, t, sigma2):
    """Literal restatement of /root/reference/src/space_time/to_gauss_markov.jl:1-20 +
    rectilinear_grid.jl:59-97: A = I (x) A_t, Q = (K_r + 1e-12 I) (x) Q_t, H = I (x) H_t', x0.P = K_r (x) P_t,
    vector observations (SmallOutputLGC) with Diagonal noise. `sigma2`: scalar or (T, Nr). kind == 'small'."""
    from . import dense_gp as dg
    r = np.asarray(r, dtype=np.float64)
    Nr = len(r)
    Kr = dg.kernelmatrix(k_space, r)
    A_t, a_t, Q_t, H_t, h_t, (m_t, P_t) = lgssm_components(k_time, t)
    T = n_times(t)
    ident = np.eye(Nr)
    A = np.stack([np.kron(ident, Ai) for Ai in A_t])
    a = np.stack([np.tile(ai, Nr) for ai in a_t])
    Q = np.stack([np.kron(Kr + 1e-12 * ident, Qi) for Qi in Q_t])
    H = np.stack([np.kron(ident, Hi[None, :]) for Hi in H_t])            # (n, Nr, Nr*d_t)
    h = np.stack([np.full(Nr, hi) for hi in np.atleast_1d(h_t)])
    s = np.asarray(sigma2, dtype=np.float64)
    R = (s * ident)[None] if s.ndim == 0 else np.stack([np.diag(v) for v in s.reshape(T, Nr)])
    return dict(ordering="F", kind="small", T=T, A=A, a=a, Q=Q, H=H, h=h, R=R,
                x0m=np.tile(m_t, Nr), x0P=np.kron(Kr, P_t))


# ------------------------------------------------------------------ pseudo-point approximation (space_time/pseudo_point.jl)
def _blkdiag(blocks):
    n, m = sum(b.shape[0] for b in blocks), sum(b.shape[1] for b in blocks)
    out = np.zeros((n, m))
    i = j = 0
    for b in blocks:
        out[i:i + b.shape[0], j:j + b.shape[1]] = b
        i, j = i + b.shape[0], j + b.shape[1]
    return out


def dtc_components(terms, z, r, t, jitter=1e-12):
    """Literal restatement of lgssm_components(::DTCSeparable, ::RectilinearGrid) (pseudo_point.jl:107-144) for
    k = sum_i s_i * Separable(k_space_i, k_time_i) with `terms` = [(s_i, k_space_i, k_time_i)]: the ScaledKernel
    rule scales H_t (lti_sde.jl:344-346), the KernelSum rule stacks the components (lti_sde.jl:404-436).
    Returns A, a, Q (n, D, D), the projection (Hb (n, Mtot, D), hb), the fan-out matrix C' (N, Mtot) and x0."""
    from . import dense_gp as dg
    z, r = np.asarray(z, dtype=np.float64), np.asarray(r, dtype=np.float64)
    M = len(z)
    ident = np.eye(M)
    parts = []
    for s, k_space, k_time in terms:
        A_t, a_t, Q_t, H_t, h_t, (m_t, P_t) = lgssm_components(k_time, t)
        Kz = dg.kernelmatrix(k_space, z)
        Kzx = dg.kernelmatrix(k_space, z, r)
        C = np.linalg.solve(Kz + jitter * ident, Kzx)                       # cholesky(K_z + 1e-12 I) \ K_zx   (M, N)
        sig = np.sqrt(s)
        parts.append(dict(
            A=np.stack([np.kron(ident, Ai) for Ai in A_t]), a=np.stack([np.tile(ai, M) for ai in a_t]),
            Q=np.stack([np.kron(Kz, Qi) for Qi in Q_t]),
            Hb=np.stack([sig * np.kron(ident, Hi[None, :]) for Hi in H_t]),  # adjoint of kron(I_M, H_t)  (M, M d_t)
            C=C, m=np.tile(m_t, M), P=np.kron(Kz, P_t)))
    n = max(p["A"].shape[0] for p in parts)
    ex = lambda arr: arr if arr.shape[0] == n else np.repeat(arr, n, axis=0)
    A = np.stack([_blkdiag([ex(p["A"])[i] for p in parts]) for i in range(n)])
    a = np.stack([np.concatenate([ex(p["a"])[i] for p in parts]) for i in range(n)])
    Q = np.stack([_blkdiag([ex(p["Q"])[i] for p in parts]) for i in range(n)])
    nh = max(p["Hb"].shape[0] for p in parts)
    exh = lambda arr: arr if arr.shape[0] == nh else np.repeat(arr, nh, axis=0)
    Hb = np.stack([_blkdiag([exh(p["Hb"])[i] for p in parts]) for i in range(nh)])
    Ct = np.concatenate([p["C"] for p in parts], axis=0).T                  # map(vcat, Cs...) then adjoint: (N, Mtot)
    x0m = np.concatenate([p["m"] for p in parts])
    x0P = _blkdiag([p["P"] for p in parts])
    return A, a, Q, Hb, np.zeros((1, Hb.shape[1])), Ct, x0m, x0P


def build_lgssm_dtc(terms, z, r, t, sigma2):
    """build_lgssm(dtcify(z, fx)) (pseudo_point.jl:33, 187-196): BottleneckLGC emissions whose fan-out is a LargeOutputLGC
    (C', c = 0, Diagonal noise). kind == 'bottleneck' (oracle/lgssm_ref.py)."""
    A, a, Q, Hb, hb, Ct, x0m, x0P = dtc_components(terms, z, r, t)
    T, N = n_times(t), len(r)
    s = np.asarray(sigma2, dtype=np.float64)
    R = (s * np.eye(N))[None] if s.ndim == 0 else np.stack([np.diag(v) for v in s.reshape(T, N)])
    return dict(ordering="F", kind="bottleneck", T=T, A=A, a=a, Q=Q, Hb=Hb, hb=hb, H=Ct[None], h=np.zeros((1, N)), R=R,
                x0m=x0m, x0P=x0P)


def dtc_kernel_diagonals(terms, r, t):
    """kernel_diagonals (pseudo_point.jl:83-105): prior variances k((r_p, t_q), (r_p, t_q)), shape (T, N)."""
    from . import dense_gp as dg
    r = np.asarray(r, dtype=np.float64)
    T = n_times(t)
    out = np.zeros((T, len(r)))
    for s, k_space, k_time in terms:
        out += s * np.outer(np.full(T, float(dg.kappa(k_time, np.zeros(1))[0])), dg.kappa(k_space, np.zeros(len(r))))
    return out


def dtc_statespace(terms, z, r, t, sigma2, y, missing=None):
    """dtc(fx, y, z_r) = logpdf(dtcify(z_r, fx), y) (pseudo_point.jl:52-54)."""
    from . import lgssm_ref as ref
    model = build_lgssm_dtc(terms, z, r, t, sigma2)
    Y = np.asarray(y, dtype=np.float64).reshape(model["T"], len(r))
    if missing is None:
        return ref.logpdf(model, Y)
    return ref.logpdf_missing(model, Y, np.asarray(missing, dtype=bool).reshape(Y.shape))


def elbo_statespace(terms, z, r, t, sigma2, y):
    """elbo(fx, y, z_r) (pseudo_point.jl:61-81) without missing data."""
    from . import lgssm_ref as ref
    model = build_lgssm_dtc(terms, z, r, t, sigma2)
    T, N = model["T"], len(r)
    Y = np.asarray(y, dtype=np.float64).reshape(T, N)
    _, covs = ref.marginals(model)                                           # marginals_diag: emission marginals incl. noise
    Cf = dtc_kernel_diagonals(terms, r, t)
    tmp = 0.0
    for q in range(T):
        Sig = np.diag(ref._at(model["R"], q))
        tmp += np.sum((Cf[q] - np.diag(covs[q])) / Sig) + N                  # sum(diag(S \ (Cf - marg.P))) - 0 + size(S, 1)
    return ref.logpdf(model, Y) - tmp / 2.0
