"""GP -> LTI SDE -> LGSSM: the host-side layer that feeds the device hot path, mirroring the API surface of

    /root/reference/src/gp/lti_sde.jl            LTISDE, to_sde, FiniteLTISDE methods (marginals, mean_and_var, mean, var,
                                                 rand, logpdf), build_lgssm, lgssm_components, kernel -> SDE tables
    /root/reference/src/gp/posterior_lti_sde.jl  PosteriorLTISDE: marginals / mean_and_var / rand / logpdf of the posterior
    /root/reference/src/util/regular_data.jl     RegularSpacing
    /root/reference/src/util/storage_types.jl    StorageType tags -- HIPStorage is the new tag that selects this backend

This layer is O(#distinct dt) host work for regular spacing (one matrix exponential) exactly as in the
reference; every T-step recursion it triggers runs on the GPU through `lgssm.py` -> libtgp_hip.so.
Kernels are small classes (KernelFunctions.jl names); `sigma2 * k`, `k1 + k2`, `k1 * k2` and
`k.stretch(s)` (== k ∘ ScaleTransform(s)) build the algebra.
"""
import numpy as np
from scipy.linalg import block_diag, expm
from scipy.special import ive

from . import lgssm as L


# ------------------------------------------------------------------------------------------ storage tag / inputs
class HIPStorage:
    """StorageType tag selecting the MI355X backend (the seam of storage_types.jl:1-61)."""

    def __init__(self, eltype=np.float64, device=0):
        if np.dtype(eltype) != np.float64:
            raise ValueError("the HIP backend computes in Float64")
        self.eltype, self.device = np.float64, device


class RegularSpacing:
    """RegularSpacing(t0, dt, N) (regular_data.jl:8-22)."""

    def __init__(self, t0, dt, N):
        self.t0, self.dt, self.N = float(t0), float(dt), int(N)

    def __len__(self):
        return self.N

    def __getitem__(self, n):
        return self.t0 + n * self.dt

    def step(self):
        return self.dt

    def collect(self):
        return self.t0 + np.arange(self.N) * self.dt

    def __eq__(self, other):
        return isinstance(other, RegularSpacing) and (self.t0, self.dt, self.N) == (other.t0, other.dt, other.N)


def _times(x):
    return x.collect() if isinstance(x, RegularSpacing) else np.asarray(x, dtype=np.float64)


def _same_inputs(x1, x2):
    if isinstance(x1, RegularSpacing) and isinstance(x2, RegularSpacing):
        return x1 == x2
    a, b = _times(x1), _times(x2)
    return a.shape == b.shape and np.array_equal(a, b)


def _same_sorted_inputs(x1, x2):
    """the same inputs AND in time order: what the shortcuts of `rand` / `logpdf` of a posterior need -- the reference's chain
    (posterior_lti_sde.jl:48-78) sorts the joined inputs (merge_datasets), the shortcuts bind the training inputs as they stand (round-5 advice)"""
    if not _same_inputs(x1, x2):
        return False
    if isinstance(x1, RegularSpacing):
        return x1.dt >= 0
    t = _times(x1)
    return bool(np.all(t[1:] >= t[:-1]))


# ------------------------------------------------------------------------------------------ kernels
class Kernel:
    def __add__(self, other):
        return KernelSum(*(self._terms(KernelSum) + other._terms(KernelSum)))

    def __mul__(self, other):
        if isinstance(other, Kernel):
            return KernelProduct(*(self._terms(KernelProduct) + other._terms(KernelProduct)))
        return ScaledKernel(float(other), self)

    __rmul__ = __mul__

    def _terms(self, cls):
        return list(self.kernels) if isinstance(self, cls) else [self]

    def stretch(self, s):
        """k ∘ ScaleTransform(s)."""
        return StretchedKernel(float(s), self)


def _cm(n, vals):
    return np.array(vals, dtype=np.float64).reshape(n, n, order="F")    # SMatrix{n,n}(column-major literals)


class SimpleKernel(Kernel):
    """to_sde -> (F, q, H); stationary_distribution -> (m, P)."""

    def sde_blocks(self):
        """(F, H, m0, P0) of the ONE LTI SDE the whole kernel expression corresponds to."""
        F, _, H = self.to_sde()
        m, P = self.stationary_distribution()
        return F, H, m, P

    def lgssm_components(self, t):
        x0 = self.stationary_distribution()
        A, a, Q, H, h = broadcast_components(self.to_sde(), x0, t)
        return A, a, Q, H, h, x0


class Matern12Kernel(SimpleKernel):      # lti_sde.jl:189-203
    def to_sde(self):
        return np.array([[-1.0]]), 2.0, np.array([1.0])

    def stationary_distribution(self):
        return np.zeros(1), np.array([[1.0]])


class Matern32Kernel(SimpleKernel):      # lti_sde.jl:205-218
    def to_sde(self):
        lam = np.sqrt(3.0)
        return _cm(2, [0, -3, 1, -2 * lam]), 4 * lam ** 3, np.array([1.0, 0.0])

    def stationary_distribution(self):
        return np.zeros(2), np.diag([1.0, 3.0])


class Matern52Kernel(SimpleKernel):      # lti_sde.jl:222-235
    def to_sde(self):
        lam = np.sqrt(5.0)
        return _cm(3, [0, 0, -lam ** 3, 1, 0, -3 * lam ** 2, 0, 1, -3 * lam]), 8 * lam ** 5 / 3, np.array([1.0, 0.0, 0.0])

    def stationary_distribution(self):
        kap = 5.0 / 3.0
        return np.zeros(3), _cm(3, [1, 0, -kap, 0, kap, 0, -kap, 0, 25])


class CosineKernel(SimpleKernel):        # lti_sde.jl:239-252
    def to_sde(self):
        return _cm(2, [0, 1, -1, 0]), 0.0, np.array([1.0, 0.0])

    def stationary_distribution(self):
        return np.zeros(2), np.eye(2)


class ConstantKernel(SimpleKernel):      # lti_sde.jl:311-321
    def __init__(self, c=1.0):
        self.c = float(c)

    def to_sde(self):
        return np.array([[0.0]]), 0.0, np.array([1.0])

    def stationary_distribution(self):
        return np.zeros(1), np.array([[self.c]])


class ApproxPeriodicKernel(SimpleKernel):  # lti_sde.jl:255-307 (sum of N cosine kernels; default N = 7)
    def __init__(self, N=7, r=1.0):
        self.N, self.r = int(N), float(r)

    def to_sde(self):
        F0, _, H0 = CosineKernel().to_sde()
        return block_diag(*[2 * np.pi * i * F0 for i in range(self.N)]), 0.0, np.tile(H0, self.N)

    def stationary_distribution(self):
        l2 = 1.0 / (4.0 * self.r ** 2)
        Ps = [(1 + (j != 1)) * ive(j - 1, l2) * np.eye(2) for j in range(1, self.N + 1)]   # besseli(j-1,l2)/exp(l2)
        return np.zeros(2 * self.N), block_diag(*Ps)


class ScaledKernel(Kernel):              # lti_sde.jl:324-346
    def __init__(self, sigma2, kernel):
        self.sigma2, self.kernel = sigma2, kernel

    def to_sde(self):
        F, q, H = self.kernel.to_sde()
        return F, self.sigma2 * q, np.sqrt(self.sigma2) * H

    def stationary_distribution(self):
        return self.kernel.stationary_distribution()

    def lgssm_components(self, t):
        A, a, Q, H, h, x0 = self.kernel.lgssm_components(t)
        sig = np.sqrt(self.sigma2)
        return A, a, Q, sig * H, sig * h, x0

    def sde_blocks(self):
        F, H, m, P = self.kernel.sde_blocks()
        return F, np.sqrt(self.sigma2) * H, m, P


class StretchedKernel(Kernel):           # lti_sde.jl:350-373
    def __init__(self, s, kernel):
        self.s, self.kernel = s, kernel

    def to_sde(self):
        F, q, H = self.kernel.to_sde()
        return F * self.s, q, H

    def stationary_distribution(self):
        return self.kernel.stationary_distribution()

    def lgssm_components(self, t):
        if isinstance(t, RegularSpacing):
            t2 = RegularSpacing(self.s * t.t0, self.s * t.dt, t.N)
        else:
            t2 = self.s * np.asarray(t, dtype=np.float64)
        return self.kernel.lgssm_components(t2)

    def sde_blocks(self):
        F, H, m, P = self.kernel.sde_blocks()
        return F * self.s, H, m, P          # exp(F s dt) == the inner kernel on stretched inputs


class KernelProduct(Kernel):             # lti_sde.jl:377-400
    def __init__(self, *kernels):
        self.kernels = kernels

    def lgssm_components(self, t):
        for k in self.kernels:       # the reference multiplies the factors' SDEs (to_sde.(k.kernels)): sums and products have no to_sde method there
            if not hasattr(k, "to_sde"):
                raise TypeError(f"a factor of a KernelProduct must have an SDE form (to_sde); {type(k).__name__} has none (lti_sde.jl:377-400)")
        sdes = [k.to_sde() for k in self.kernels]
        x0s = [k.stationary_distribution() for k in self.kernels]
        F, H, m0, P0 = sdes[0][0], sdes[0][2], x0s[0][0], x0s[0][1]
        for (Fb, _, Hb), (mb, Pb) in zip(sdes[1:], x0s[1:]):
            F = np.kron(F, np.eye(Fb.shape[0])) + np.kron(np.eye(F.shape[0]), Fb)
            H, m0, P0 = np.kron(H, Hb), np.kron(m0, mb), np.kron(P0, Pb)
        q = float(np.prod([s[1] for s in sdes]))
        A, a, Q, Hs, hs = broadcast_components((F, q, H), (m0, P0), t)
        return A, a, Q, Hs, hs, (m0, P0)

    def sde_blocks(self):
        parts = [k.sde_blocks() for k in self.kernels]
        F, H, m0, P0 = parts[0]
        for Fb, Hb, mb, Pb in parts[1:]:
            F = np.kron(F, np.eye(Fb.shape[0])) + np.kron(np.eye(F.shape[0]), Fb)
            H, m0, P0 = np.kron(H, Hb), np.kron(m0, mb), np.kron(P0, Pb)
        return F, H, m0, P0


class KernelSum(Kernel):                 # lti_sde.jl:404-445
    def __init__(self, *kernels):
        self.kernels = kernels

    def lgssm_components(self, t):
        parts = [k.lgssm_components(t) for k in self.kernels]
        T = len(t)

        def stack(idx, join):
            n = max(p[idx].shape[0] for p in parts)
            ex = [p[idx] if p[idx].shape[0] == n else np.repeat(p[idx], n, axis=0) for p in parts]
            return np.stack([join([e[i] for e in ex]) for i in range(n)])
        A = stack(0, lambda bs: block_diag(*bs))
        Q = stack(2, lambda bs: block_diag(*bs))
        a = stack(1, np.concatenate)
        H = stack(3, np.concatenate)
        h = stack(4, lambda bs: np.sum(bs))
        m0 = np.concatenate([p[5][0] for p in parts])
        P0 = block_diag(*[p[5][1] for p in parts])
        assert A.shape[0] in (1, T)
        return A, a, Q, H, h, (m0, P0)

    def sde_blocks(self):
        parts = [k.sde_blocks() for k in self.kernels]
        return (block_diag(*[p[0] for p in parts]), np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts]),
                block_diag(*[p[3] for p in parts]))


def broadcast_components(FqH, x0, t):
    """lti_sde.jl:135-160: A = exp(F dt), Q = P - A P A'. Regular spacing -> one shared block (Fill);
    irregular -> per-step blocks with dt_1 = 1 (t0 := t1 - 1, lti_sde.jl:139)."""
    F, _, H = FqH
    _, P0 = x0
    P = _symmetric_from_upper(P0)
    d = F.shape[0]
    if isinstance(t, RegularSpacing):
        A = _expm_small(np.asarray(F, dtype=np.float64) * t.dt)      # (closed form for a Matern block, scipy's expm otherwise)
        return A[None], np.zeros((1, d)), (P - A @ P @ A.T)[None], np.array(H, dtype=np.float64)[None], np.zeros(1)
    tt = np.asarray(t, dtype=np.float64)
    dts = np.diff(np.concatenate([[tt[0] - 1.0], tt]))
    uniq, inv = np.unique(dts, return_inverse=True)          # one exponential per distinct dt ...
    Au = expm_batch(F, uniq)                                  # ... all of them in one vectorised pass
    Qu = P[None] - Au @ P[None] @ np.swapaxes(Au, -1, -2)
    return Au[inv], np.zeros((1, d)), Qu[inv], np.array(H, dtype=np.float64)[None], np.zeros(1)


def expm_batch(F, dts):
    """exp(F * dt) for a whole vector of dt at once (scaling and squaring around a degree-18 Taylor polynomial,
    vectorised over the batch): irregular spacing -- every posterior query at new inputs -- needs T of them
    (lti_sde.jl:140), and one scipy call per step is the host bottleneck at T ~ 1e6."""
    dts = np.asarray(dts, dtype=np.float64)
    if len(dts) <= 4:
        return np.stack([expm(F * dt) for dt in dts])
    d = F.shape[0]
    nF = max(np.linalg.norm(F, 1), 1e-300)
    svec = np.maximum(0, np.ceil(np.log2(np.maximum(nF * np.abs(dts), 1e-300) / 0.25))).astype(int)
    out = np.empty((len(dts), d, d))
    for sv in np.unique(svec):                      # one vectorised pass per squaring depth (<= ~20 groups)
        idx = np.nonzero(svec == sv)[0]
        X = F[None] * (dts[idx] / 2.0 ** sv)[:, None, None]
        term = np.broadcast_to(np.eye(d), X.shape).copy()
        acc = term.copy()
        for k in range(1, 19):
            term = term @ X / k
            acc = acc + term
        for _ in range(sv):
            acc = acc @ acc
        out[idx] = acc
    return out


def to_kernel(spec):
    """Nested-tuple spec (as used by the test-suite / bench) -> kernel object."""
    if isinstance(spec, Kernel):
        return spec
    name = spec[0]
    simple = {"matern12": Matern12Kernel, "matern32": Matern32Kernel, "matern52": Matern52Kernel, "cosine": CosineKernel}
    if name in simple:
        return simple[name]()
    if name == "constant":
        return ConstantKernel(spec[1])
    if name == "approx_periodic":
        return ApproxPeriodicKernel(spec[1], spec[2])
    if name == "scaled":
        return ScaledKernel(float(spec[1]), to_kernel(spec[2]))
    if name == "stretched":
        return StretchedKernel(float(spec[1]), to_kernel(spec[2]))
    if name == "sum":
        return KernelSum(*[to_kernel(s) for s in spec[1:]])
    if name == "product":
        return KernelProduct(*[to_kernel(s) for s in spec[1:]])
    raise ValueError(name)


# ------------------------------------------------------------------------------------------ mean functions / GP
class ZeroMean:
    def __call__(self, t):
        return None


class ConstMean:
    def __init__(self, c):
        self.c = float(c)

    def __call__(self, t):
        return np.full(len(t), self.c)


class CustomMean:
    def __init__(self, f):
        self.f = f

    def __call__(self, t):
        return np.array([self.f(v) for v in _times(t)], dtype=np.float64)


class GP:
    def __init__(self, *args):
        if len(args) == 1:
            self.mean, self.kernel = ZeroMean(), args[0]
        else:
            m, self.kernel = args
            self.mean = ConstMean(m) if np.isscalar(m) else m


DEVICE_COMPONENTS_MIN_T = 2048      # irregular spacing: from this length on, exp(F dt) is evaluated on the device


def build_lgssm(kernel, x, sigma2s, mean=None, device=0, force_per_step=False, device_components=None):
    """lti_sde.jl:71-80 -> device LGSSM. `sigma2s`: scalar (Fill) or length-T array.
    force_per_step replicates Fill blocks to per-step arrays (the general layout, for benchmarking it).
    Irregular spacing with T >= DEVICE_COMPONENTS_MIN_T (or device_components=True): the per-step A_k = exp(F dt_k),
    Q_k are built on the device from the time stamps (SDETransitions) instead of on the host."""
    T = len(x)
    if device_components is None:
        device_components = (not isinstance(x, RegularSpacing)) and T >= DEVICE_COMPONENTS_MIN_T and not force_per_step
    if device_components and not isinstance(x, RegularSpacing) and hasattr(kernel, "sde_blocks"):
        F, H, m0, P0 = kernel.sde_blocks()
        if F.shape[0] <= 8:
            hh = np.zeros(1)
            if isinstance(mean, ConstMean):
                hh = hh + mean.c
            elif mean is not None and not isinstance(mean, ZeroMean):
                hh = np.asarray(mean(x), dtype=np.float64)
            R = np.atleast_1d(np.asarray(sigma2s, dtype=np.float64))
            A1, _, Q1, _, _, _ = kernel.lgssm_components(_times(x)[:1])       # the reference's own first step (dt_1 rule)
            trans = L.SDETransitions(L.Forward, F, _times(x), L.Gaussian(np.asarray(m0, float), np.asarray(P0, float)), A1[0], Q1[0])
            return L.LGSSM(trans, L.ScalarOutputLGC(np.asarray(H, float)[None], hh, R), T=T, device=device)
    A, a, Q, H, h, (m0, P0) = kernel.lgssm_components(x)
    mv = None if mean is None else mean(x)
    if isinstance(mean, ConstMean):
        h = h + mean.c                      # same values as hs .+ m (lti_sde.jl:126-127), but a Fill stays a Fill
    elif mv is not None:
        h = (h if h.shape[0] == T else np.repeat(h, T, axis=0)) + mv          # lti_sde.jl:118-131
    R = np.atleast_1d(np.asarray(sigma2s, dtype=np.float64))
    if force_per_step:
        rep = lambda z: z if z.shape[0] == T else np.repeat(z, T, axis=0)
        A, a, Q, H, h, R = rep(A), rep(a), rep(Q), rep(H), rep(h), rep(R)
    trans = L.GaussMarkovModel(L.Forward, A, a, Q, L.Gaussian(np.asarray(m0, dtype=np.float64), np.asarray(P0, dtype=np.float64)))
    return L.LGSSM(trans, L.ScalarOutputLGC(H, h, R), T=T, device=device)


class LTISDE:
    """lti_sde.jl:7-16."""

    def __init__(self, f, storage):
        self.f, self.storage = f, storage

    def __call__(self, x, sigma2=1e-12):            # FiniteGP(f, x, 1e-12) default jitter: lti_sde.jl:27-29
        return FiniteLTISDE(self, x, sigma2)


def to_sde(f, storage=None):
    return LTISDE(f, storage or HIPStorage())


class FiniteLTISDE:
    """FiniteGP{<:LTISDE}: the object logpdf / marginals / rand / posterior are called on."""

    def __init__(self, f, x, sigma2):
        self.f, self.x = f, x
        s = np.atleast_1d(np.asarray(sigma2, dtype=np.float64))
        if s.shape[0] not in (1, len(x)):
            raise ValueError("noise variances must be a scalar or one per input")
        self.sigma2 = s
        self._model = None
        self._model_key = None

    def _param_key(self):
        """values of everything the cached device model was built from: kernel hyper-parameters (mutable through the
        owner / attribute handles `parameters` returns), a ConstMean's constant, the noise variances"""
        vals = [float(getattr(o, a)) for _, o, a in parameters(self.f.f.kernel)]
        if isinstance(self.f.f.mean, ConstMean):
            vals.append(float(self.f.f.mean.c))
        return (tuple(vals), self.sigma2.tobytes())

    def build_lgssm(self):
        key = self._param_key()
        if self._model is None or self._model_key != key:      # parameters changed in place since the model was bound
            self._model = build_lgssm(self.f.f.kernel, self.x, self.sigma2, self.f.f.mean, self.f.storage.device)
            self._model_key = key
        return self._model


def marginals(fx):
    """-> (mean, std): the Normal marginals (lti_sde.jl:33-35 / posterior_lti_sde.jl:18-37, gaussian.jl:61-63)."""
    m, v = mean_and_var(fx)
    return m, np.sqrt(v)


def mean_and_var(fx):
    if isinstance(fx, FinitePosteriorLTISDE):
        return fx._mean_and_var()
    return L.marginals(fx.build_lgssm())


def mean(fx):
    return mean_and_var(fx)[0]


def var(fx):
    return mean_and_var(fx)[1]


def logpdf(fx, y):
    """lti_sde.jl:60-68 / posterior_lti_sde.jl:62-78. NaN entries of y are `missing`."""
    if isinstance(fx, FinitePosteriorLTISDE):
        return fx._logpdf(y)
    return L.logpdf(fx.build_lgssm(), y)


def rand(rng, fx, N=None):
    """lti_sde.jl:48-58 / posterior_lti_sde.jl:48-58."""
    if N is not None:
        return np.stack([rand(rng, fx) for _ in range(N)], axis=1)
    if isinstance(fx, FinitePosteriorLTISDE):
        return fx._rand(rng)
    return L.rand(rng, fx.build_lgssm())


def posterior(fx, y):
    """posterior_lti_sde.jl:7-10: lazy -- all the work happens in marginals / rand / logpdf of the result."""
    return PosteriorLTISDE(fx.f, dict(y=np.asarray(y, dtype=np.float64), x=fx.x, sigma2=fx.sigma2))


class PosteriorLTISDE:
    def __init__(self, prior, data):
        self.prior, self.data = prior, data

    def __call__(self, x, sigma2=1e-18):            # AbstractGPs' FiniteGP default jitter for posterior queries
        return FinitePosteriorLTISDE(self, x, sigma2)


def _noise(s, n):
    s = np.atleast_1d(np.asarray(s, dtype=np.float64))
    return s if s.shape[0] == n else np.full(n, s[0])


_pair_statistic = L._pair_statistic


class FinitePosteriorLTISDE:
    LARGE_VAR = 1e15                                 # missings.jl:43

    def __init__(self, f, x, sigma2):
        self.f, self.x = f, x
        self.sigma2 = np.atleast_1d(np.asarray(sigma2, dtype=np.float64))

    def _merge(self, sigma2_pr, y_pr_missing=True):
        """merge_datasets (posterior_lti_sde.jl:97-123): join + sort training and prediction inputs."""
        d = self.f.data
        xt, xp = _times(d["x"]), _times(self.x)
        ntr, npr = len(xt), len(xp)
        if ntr and npr and np.all(xt[1:] >= xt[:-1]) and np.all(xp[1:] >= xp[:-1]):
            # both already in temporal order (the usual case): the stable sort of the joined inputs is a MERGE -- a prediction input goes behind
            # every training input that is not later than it (ties keep the joined order: training first).  O(n) instead of two argsorts of the
            # joined array (a third of a prediction call's wall clock at a million inputs)
            pos = np.searchsorted(xt, xp, side="right")
            pr = pos + np.arange(npr)
            tr = np.arange(ntr) + np.cumsum(np.bincount(pos, minlength=ntr + 1))[:ntr]
            n = ntr + npr
            x, S, y = np.empty(n), np.empty(n), np.empty(n)
            x[tr], x[pr] = xt, xp
            S[tr], S[pr] = _noise(d["sigma2"], ntr), sigma2_pr
            y[tr], y[pr] = d["y"], np.nan
            return x, S, y, tr, pr
        x_raw = np.concatenate([xt, xp])
        order = np.argsort(x_raw, kind="stable")
        inv = np.argsort(order, kind="stable")
        S = np.concatenate([_noise(d["sigma2"], ntr), sigma2_pr])[order]
        y = np.concatenate([d["y"], np.full(npr, np.nan)])[order]
        return x_raw[order], S, y, inv[:ntr], inv[ntr:]

    def _posterior_model(self, x, S, y):
        prior = self.f.prior
        return build_lgssm(prior.f.kernel, x, S, prior.f.mean, prior.storage.device)

    def _mean_and_var(self):
        d = self.f.data
        # the reference's own call chain (posterior_lti_sde.jl:18-36): posterior(model, ys) is lazy on this backend and
        # replace_observation_noise_cov only records the new noise, so marginals(...) is ONE fused filter + RTS smoother
        if _same_inputs(self.x, d["x"]):             # :27-36 -- the benchmarked path
            model = self._posterior_model(d["x"], d["sigma2"], d["y"])
            S_new = _noise(self.sigma2, len(self.x)) if len(self.sigma2) > 1 else self.sigma2
            return L.marginals(L.replace_observation_noise_cov(L.posterior(model, d["y"]), S_new))
        npr = len(self.x)
        x, S, y, _, pr = self._merge(np.full(npr, self.LARGE_VAR))
        s_full = np.zeros(len(x))
        s_full[pr] = _noise(self.sigma2, npr)        # build_prediction_obs_vars :136-144
        m, v = L.marginals(L.replace_observation_noise_cov(L.posterior(self._posterior_model(x, S, y), y), s_full))
        return m[pr], v[pr]

    def _rand(self, rng):
        d = self.f.data
        if _same_sorted_inputs(self.x, d["x"]) and not np.isnan(d["y"]).any():
            # prediction inputs = training inputs: each joined pair of steps shares ONE latent state (dt = 0: A = I, Q = 0), so the draw at
            # the prediction step is h'x_t + sqrt(s_t) eps of the posterior's reverse-time chain over the T training steps -- the posterior
            # with its observation noise replaced, which is one launch on the prior's stationary structure (tgp_posterior_rand) instead
            # of 2T steps of the general engine.  Same distribution; the draws of one rng are consumed in a different order
            model = self._posterior_model(d["x"], d["sigma2"], d["y"])
            S_new = _noise(self.sigma2, len(self.x)) if len(self.sigma2) > 1 else self.sigma2
            return L.rand(rng, L.replace_observation_noise_cov(L.posterior(model, d["y"]), S_new))
        return self._rand_merged(rng)

    def _rand_merged(self, rng):
        npr = len(self.x)
        x, S, y, _, pr = self._merge(np.full(npr, self.LARGE_VAR))
        s_full = np.zeros(len(x))
        s_full[pr] = _noise(self.sigma2, npr)
        post = L.replace_observation_noise_cov(L.posterior(self._posterior_model(x, S, y), y), s_full)
        return L.rand(rng, post)[pr]

    def _logpdf(self, y_pr):
        d = self.f.data
        y_pr = np.asarray(y_pr, dtype=np.float64)
        if y_pr.shape != (len(self.x),):
            raise ValueError("Dimension mismatch: one observation per prediction input")
        if _same_sorted_inputs(self.x, d["x"]):
            # prediction inputs = training inputs: every joined pair of steps is ONE latent value observed twice (dt = 0: A = I, Q = 0), so the
            # posterior over the T training steps with its noise replaced IS the model the chain below would build over 2T steps -- and its logpdf
            # needs no posterior: log p(y* | y) = log p(ybar) + pair - log p(y) (lgssm.py `_posterior_logpdf_pair`), two logpdf calls on the prior's
            # own handle for an LTI model instead of the evaluated posterior over 2T joined steps
            model = self._posterior_model(d["x"], d["sigma2"], d["y"])
            S_new = _noise(self.sigma2, len(self.x)) if len(self.sigma2) > 1 else self.sigma2
            return L.logpdf(L.replace_observation_noise_cov(L.posterior(model, d["y"]), S_new), y_pr)
        return self._logpdf_joint(y_pr)

    def _logpdf_joint(self, y_pr):
        """The reference's chain (posterior_lti_sde.jl:62-78) as it stands: posterior of the prior over merge_datasets' joined inputs with
        the prediction inputs missing, the noise replaced, logpdf of the prediction observations with the training inputs missing.  The
        posterior is lazy and its logpdf is log p(y, y*) - log p(y) (lgssm.py `_posterior_logpdf_pair`: every joined step is observed on
        exactly one side): the PRIOR's logpdf over the joined inputs with both data sets in place, minus the prior's logpdf of the
        training data over the same steps -- no reverse-time model of T x (2 d^2 + d) doubles evaluated or filtered."""
        npr = len(self.x)
        s_pr = _noise(self.sigma2, npr)
        x, S, y, tr, pr = self._merge(s_pr)
        s_full = np.zeros(len(x))
        s_full[pr] = s_pr
        post = L.replace_observation_noise_cov(L.posterior(self._posterior_model(x, S, y), y), s_full)
        y_full = np.full(len(x), np.nan)             # build_prediction_obs :148-158: training points are missing
        y_full[pr] = y_pr
        return L.logpdf(post, y_full)

    def _logpdf_merged(self, y_pr):
        npr = len(self.x)
        s_pr = _noise(self.sigma2, npr)
        x, S, y, tr, pr = self._merge(s_pr)
        s_full = np.zeros(len(x))
        s_full[pr] = s_pr
        post = L.replace_observation_noise_cov(L.posterior(self._posterior_model(x, S, y), y), s_full)
        y_full = np.full(len(x), np.nan)             # build_prediction_obs :148-158: training points are missing
        y_full[pr] = y_pr
        return L.logpdf(post.materialise(), y_full)  # (the EVALUATED posterior: what the chain costs without the lazy object; tests hold the two together)


# ------------------------------------------------------------------------------------------ gradient of logpdf
def parameters(kernel, prefix="kernel"):
    """Flat list of (name, owner, attribute) for the positive hyper-parameters of a kernel expression."""
    out = []
    if isinstance(kernel, ScaledKernel):
        out.append((prefix + ".sigma2", kernel, "sigma2"))
        out += parameters(kernel.kernel, prefix + ".kernel")
    elif isinstance(kernel, StretchedKernel):
        out.append((prefix + ".s", kernel, "s"))
        out += parameters(kernel.kernel, prefix + ".kernel")
    elif isinstance(kernel, ConstantKernel):
        out.append((prefix + ".c", kernel, "c"))
    elif isinstance(kernel, ApproxPeriodicKernel):
        out.append((prefix + ".r", kernel, "r"))
    elif isinstance(kernel, (KernelSum, KernelProduct)):
        for i, k in enumerate(kernel.kernels):
            out += parameters(k, f"{prefix}.kernels[{i}]")
    return out


# ---- exact derivatives of the O(1) host map hyper-parameter -> shared blocks (regular spacing) ---------------------------------------
_EYE = {}


def _expm_small(X):
    """exp(X) of a small matrix.  X = -lam I + N with N nilpotent -- every Matern block's F dt, stretched or not -- has
    exp(X) = e^(-lam) (I + N + N^2 / 2 + ...) in closed form (what the device evaluates per step on irregular inputs): a few
    products of 3 x 3 matrices against scipy's Pade approximant with its norm estimates (~30 us a piece: a quarter of the host's share of a
    gradient evaluation).  Anything else goes to scipy.linalg.expm."""
    n = X.shape[0]
    if n <= 4 and X.shape[1] == n:
        I = _EYE.get(n)
        if I is None:
            I = _EYE[n] = np.eye(n)
        lam = -float(np.trace(X)) / n
        N = X + lam * I
        S, Nk, f = I + N, N, 1.0
        for k in range(2, n):
            Nk = Nk @ N
            f *= k
            S = S + Nk / f
        Nn = Nk @ N if n > 1 else N
        # the series stops at N^(n-1): what it leaves out is N^n / n! + N^(n+1) / (n+1)! + ...  Accepted only when its first two terms are below
        # the rounding of exp(X)'s own entries -- an ABSOLUTE bound tied to S, not a ratio to max|N|^n (round-5 advice: a block-diagonal sum
        # of two nearly equal stiff blocks can pass a ratio test while N is not nilpotent); beyond them the terms fall by |N| / k or, for a
        # structurally nilpotent N, are rounding residue of the products themselves
        nn = float(np.max(np.abs(Nn))) if n > 1 else 0.0
        tol = 1e-13 * float(np.max(np.abs(S)))
        if nn == 0.0 or (nn / (f * n) <= tol and nn * float(np.max(np.sum(np.abs(N), axis=1))) / (f * n * (n + 1)) <= tol):
            return np.exp(-lam) * S
    return expm(X)


def _expm_and_tangent(X, E, cache=None):
    """exp(X) and its Frechet derivative in direction E. E = 0: no derivative; E commuting with X (every rule of `_sde_jet`: a stretch
    scales F, a product's factor enters as a Kronecker summand): dexp = E exp(X); otherwise Van Loan (the upper-right block of
    exp([[X, E], [0, X]])). `cache`: exp(X) of a kernel expression is the same for every hyper-parameter of one gradient."""
    key = None if cache is None else X.tobytes()
    A = None if key is None else cache.get(key)
    if not np.any(E):
        if A is None:
            A = _expm_small(X)
            if key is not None:
                cache[key] = A
        return A, np.zeros_like(X)
    if np.max(np.abs(X @ E - E @ X)) <= 1e-14 * max(1e-300, np.max(np.abs(X)) * np.max(np.abs(E))):
        if A is None:
            A = _expm_small(X)
            if key is not None:
                cache[key] = A
        return A, E @ A
    n = X.shape[0]
    M = np.zeros((2 * n, 2 * n))
    M[:n, :n] = M[n:, n:] = X
    M[:n, n:] = E
    W = expm(M)
    return W[:n, :n], W[:n, n:]


def _sde_jet(kernel, target):
    """(F, dF, H, dH, m0, P0, dP0): `sde_blocks()` of a kernel expression with the derivative w.r.t. ONE hyper-parameter
    target = (owner object, attribute name). Mirrors to_sde / stationary_distribution class by class."""
    if isinstance(kernel, ScaledKernel):
        F, dF, H, dH, m, P, dP = _sde_jet(kernel.kernel, target)
        sig = np.sqrt(kernel.sigma2)
        own = target[0] is kernel and target[1] == "sigma2"
        return F, dF, sig * H, sig * dH + (H / (2.0 * sig) if own else 0.0), m, P, dP
    if isinstance(kernel, StretchedKernel):
        F, dF, H, dH, m, P, dP = _sde_jet(kernel.kernel, target)
        own = target[0] is kernel and target[1] == "s"
        return F * kernel.s, dF * kernel.s + (F if own else 0.0), H, dH, m, P, dP
    if isinstance(kernel, KernelSum):
        parts = [_sde_jet(k, target) for k in kernel.kernels]
        cat = lambda i: np.concatenate([p[i] for p in parts])
        bd = lambda i: block_diag(*[p[i] for p in parts])
        return bd(0), bd(1), cat(2), cat(3), cat(4), bd(5), bd(6)
    if isinstance(kernel, KernelProduct):
        parts = [_sde_jet(k, target) for k in kernel.kernels]
        F, dF, H, dH, m, P, dP = parts[0]
        for Fb, dFb, Hb, dHb, mb, Pb, dPb in parts[1:]:
            Ia, Ib = np.eye(F.shape[0]), np.eye(Fb.shape[0])
            F, dF = np.kron(F, Ib) + np.kron(Ia, Fb), np.kron(dF, Ib) + np.kron(Ia, dFb)
            H, dH = np.kron(H, Hb), np.kron(dH, Hb) + np.kron(H, dHb)
            P, dP = np.kron(P, Pb), np.kron(dP, Pb) + np.kron(P, dPb)
            m = np.kron(m, mb)
        return F, dF, H, dH, m, P, dP
    F, _, H = kernel.to_sde()
    m, P = kernel.stationary_distribution()
    dP = np.zeros_like(P)
    if target[0] is kernel:
        if isinstance(kernel, ConstantKernel) and target[1] == "c":
            dP = np.array([[1.0]])
        elif isinstance(kernel, ApproxPeriodicKernel) and target[1] == "r":
            # P_j = (1 + [j != 1]) ive(j - 1, l2) I,  l2 = 1 / (4 r^2);  d ive(v, z) / dz = (ive(v-1, z) + ive(v+1, z)) / 2 - ive(v, z)
            l2 = 1.0 / (4.0 * kernel.r ** 2)
            dl2 = -1.0 / (2.0 * kernel.r ** 3)
            dv = lambda v: 0.5 * (ive(v - 1, l2) + ive(v + 1, l2)) - ive(v, l2)
            dP = block_diag(*[(1 + (j != 1)) * dv(j - 1) * dl2 * np.eye(2) for j in range(1, kernel.N + 1)])
        else:
            raise NotImplementedError(f"no derivative rule for {type(kernel).__name__}.{target[1]}")
    return np.asarray(F, float), np.zeros_like(np.asarray(F, float)), np.asarray(H, float), np.zeros_like(np.asarray(H, float)), m, np.asarray(P, float), dP


_UPPER = {}


def _symmetric_from_upper(M):
    """Symmetric(M): the upper triangle mirrored (np.triu twice costs ~10 us per small matrix -- ten of them per gradient evaluation)"""
    M = np.asarray(M, float)
    n = M.shape[0]
    iu = _UPPER.get(n)
    if iu is None:
        iu = _UPPER[n] = np.triu_indices(n, 1)
    S = M.copy()
    S[iu[1], iu[0]] = M[iu]
    return S


def _holds(kernel, owner, cache):
    """does the kernel expression contain the object that owns the target parameter?"""
    key = ("owners", id(kernel))
    ids = cache.get(key)
    if ids is None:
        ids = cache[key] = frozenset(id(o) for _, o, _ in parameters(kernel))
    return id(owner) in ids


def _components_jet(kernel, dt, ddt, target, first=False, cache=None):
    """Shared blocks (A, Q, H, m0, P0) of `lgssm_components(RegularSpacing(., dt, .))` and their derivatives (dA, dQ, dH, dP0)
    w.r.t. the target hyper-parameter; `ddt` is the derivative of this sub-expression's (stretched) time step.
    first=True: the FIRST transition of an irregularly spaced input (lti_sde.jl:139: dt_1 := 1 in each sub-kernel's own stretched
    time, so a ScaleTransform does not reach it -- except through the F of a product's factors, as in the reference)."""
    if isinstance(kernel, ScaledKernel):
        A, dA, Q, dQ, H, dH, m, P, dP = _components_jet(kernel.kernel, dt, ddt, target, first, cache)
        sig = np.sqrt(kernel.sigma2)
        own = target[0] is kernel and target[1] == "sigma2"
        return A, dA, Q, dQ, sig * H, sig * dH + (H / (2.0 * sig) if own else 0.0), m, P, dP
    if isinstance(kernel, StretchedKernel):
        own = target[0] is kernel and target[1] == "s"
        if first:
            return _components_jet(kernel.kernel, dt, ddt, target, True, cache)
        return _components_jet(kernel.kernel, kernel.s * dt, kernel.s * ddt + (dt if own else 0.0), target, False, cache)
    if isinstance(kernel, KernelSum):
        # a summand that does not hold the target parameter has zero derivatives and the same values for every parameter of one gradient:
        # computed once (eight parameters over three summands: 10 jets instead of 24)
        parts = []
        for k in kernel.kernels:
            if cache is not None and not _holds(k, target[0], cache):
                key = ("values", id(k), float(dt), float(ddt), bool(first))
                part = cache.get(key)
                if part is None:
                    part = cache[key] = _components_jet(k, dt, ddt, (None, ""), first, cache)
                parts.append(part)
            else:
                parts.append(_components_jet(k, dt, ddt, target, first, cache))
        bd = lambda i: block_diag(*[p[i] for p in parts])
        cat = lambda i: np.concatenate([p[i] for p in parts])
        return bd(0), bd(1), bd(2), bd(3), cat(4), cat(5), cat(6), bd(7), bd(8)
    F, dF, H, dH, m, P0, dP = _sde_jet(kernel, target)          # simple kernels and products: one SDE, one exponential
    P = _symmetric_from_upper(P0)
    dPs = _symmetric_from_upper(dP)
    A, dA = _expm_and_tangent(F * dt, dF * dt + F * ddt, cache)
    Q = P - A @ P @ A.T
    dQ = dPs - dA @ P @ A.T - A @ dPs @ A.T - A @ P @ dA.T
    return A, dA, Q, dQ, H, dH, m, P0, dP


def _shared_block_tangents(fx, names, plist):
    """exact d (A, a, Q, H, h, R, x0m, x0P) / d parameter for every name in `names` (regular spacing, homoscedastic noise)"""
    kernel, cache = fx.f.f.kernel, {}
    out = []
    for name in names:
        if name == "noise":
            out.append(dict(R=1.0))
            continue
        if name == "mean.c":
            out.append(dict(h=1.0))
            continue
        owner, attr = next((o, a) for n, o, a in plist if n == name)
        A, dA, Q, dQ, H, dH, m, P, dP = _components_jet(kernel, fx.x.dt, 0.0, (owner, attr), cache=cache)
        out.append(dict(A=dA, Q=dQ, H=dH, x0P=dP))
    return out


def _sde_param_tangents(fx, names, plist):
    """exact derivatives of the O(1) blocks of an SDE-described (irregularly spaced) model: F, H, x0P and the first transition A1, Q1"""
    kernel = fx.f.f.kernel
    out = []
    for name in names:
        if name == "noise":
            out.append(dict(R=1.0))
            continue
        if name == "mean.c":
            out.append(dict(h=1.0))
            continue
        tgt = next((o, a) for n, o, a in plist if n == name)
        _, dF, _, dH, _, _, dP = _sde_jet(kernel, tgt)
        _, dA1, _, dQ1, _, _, _, _, _ = _components_jet(kernel, 1.0, 0.0, tgt, first=True)
        out.append(dict(F=dF, H=dH, x0P=dP, A1=dA1, Q1=dQ1))
    return out


def _logpdf_and_gradient_adjoint(fx, y):
    """One adjoint pass on the device (block gradients) contracted with the exact block tangents: cost independent of the number of
    hyper-parameters."""
    # (layouts the pass does not cover raise; regular spacing with one noise variance and a zero / constant mean is all-shared by construction:
    #  evaluating the components -- a matrix exponential -- only to look at their shapes was a fifth of the host's share of a gradient)
    if not (isinstance(fx.x, RegularSpacing) and fx.sigma2.shape[0] == 1 and isinstance(fx.f.f.mean, (ZeroMean, ConstMean))):
        _shared_blocks(fx)
    plist = parameters(fx.f.f.kernel)
    names = [n for n, _, _ in plist] + ["noise"] + (["mean.c"] if isinstance(fx.f.f.mean, ConstMean) else [])
    lp, g = L.logpdf_adjoint(fx.build_lgssm(), y)
    grad = {}
    for name, t in zip(names, _shared_block_tangents(fx, names, plist)):
        grad[name] = float(sum(_contract(g[k], v) for k, v in t.items()))
    return lp, grad


def _contract(a, b):
    """sum(a * b) of two same-shaped small arrays or scalars (np.sum of a product costs ~5 us a piece: a dozen per gradient)"""
    if isinstance(a, np.ndarray) and isinstance(b, np.ndarray):
        return float(np.dot(a.ravel(), b.ravel()))
    return float(np.sum(np.asarray(a) * np.asarray(b)))


def _shared_blocks(fx):
    k, mean = fx.f.f.kernel, fx.f.f.mean
    A, a, Q, H, h, (m0, P0) = k.lgssm_components(fx.x)
    if any(z.shape[0] != 1 for z in (A, a, Q, H, h)) or fx.sigma2.shape[0] != 1:
        raise NotImplementedError("logpdf_and_gradient: regular spacing and homoscedastic noise (all blocks shared) only")
    hh = float(h[0])
    if isinstance(mean, ConstMean):
        hh += mean.c
    elif not isinstance(mean, ZeroMean):
        raise NotImplementedError("logpdf_and_gradient: ZeroMean or ConstMean only")
    return dict(A=A[0], a=a[0], Q=Q[0], H=H[0], h=hh, R=float(fx.sigma2[0]), x0m=np.asarray(m0, float), x0P=np.asarray(P0, float))


def _sde_param_blocks(fx):
    """the O(1) blocks an SDE-described (irregularly spaced) model is made of, as a function of the hyper-parameters"""
    k, mean = fx.f.f.kernel, fx.f.f.mean
    F, H, m0, P0 = k.sde_blocks()
    A1, _, Q1, _, _, _ = k.lgssm_components(_times(fx.x)[:1])
    if isinstance(mean, ConstMean):
        hh = float(mean.c)
    elif isinstance(mean, ZeroMean):
        hh = 0.0
    else:
        raise NotImplementedError("logpdf_and_gradient: ZeroMean or ConstMean only")
    out = dict(F=np.asarray(F, float), H=np.asarray(H, float), x0m=np.asarray(m0, float), x0P=np.asarray(P0, float),
               A1=np.asarray(A1[0], float), Q1=np.asarray(Q1[0], float), h=hh)
    if fx.sigma2.shape[0] == 1:
        out["R"] = float(fx.sigma2[0])
    return out


def _logpdf_and_gradient_sde(fx, y, rel_step):
    """irregular spacing: the model is bound through its SDE and the per-step tangents are formed on the device"""
    kernel = fx.f.f.kernel
    if not hasattr(kernel, "sde_blocks") or kernel.sde_blocks()[0].shape[0] > 4:
        raise NotImplementedError("logpdf_and_gradient on irregular inputs: kernels with state dimension <= 4")
    plist = parameters(kernel)
    shared_noise = fx.sigma2.shape[0] == 1
    names = [n for n, _, _ in plist] + (["noise"] if shared_noise else []) + (["mean.c"] if isinstance(fx.f.f.mean, ConstMean) else [])
    tangents = _sde_param_tangents(fx, names, plist)
    model = build_lgssm(kernel, fx.x, fx.sigma2, fx.f.f.mean, fx.f.storage.device, device_components=True)
    if not isinstance(model.transitions, L.SDETransitions):
        raise NotImplementedError("logpdf_and_gradient: could not bind the model through its SDE")
    lp, g = L.logpdf_and_grad_sde(model, y, tangents, rel_step)
    return lp, dict(zip(names, g))


def _logpdf_and_gradient_fd(fx, y, rel_step=1e-4):
    """Central differences of the device logpdf: 2 evaluations per parameter, each a re-bind of the (O(1), shared-block) model
    plus one logpdf on the group kernels. Relative accuracy ~1e-7 (the h^2 truncation term and the 1e-11 rounding of logpdf
    divided by 2 h balance around h = 1e-4 |v|). The step is RELATIVE to the parameter (never larger than half of it), so that a
    small positive parameter -- a noise variance of 1e-6 -- is never evaluated at a negative value; whatever happens in an
    evaluation, the perturbed attribute is restored."""
    # heteroscedastic noise (one variance per observation): the kernel's and the mean's parameters are differenced as always; there is no
    # single "noise" parameter, so that entry is left out (T per-step derivatives are not what a hyper-parameter search asks for)
    shared_noise = fx.sigma2.shape[0] == 1
    plist = parameters(fx.f.f.kernel)
    entries = list(plist) + ([("noise", None, None)] if shared_noise else []) + ([("mean.c", fx.f.f.mean, "c")] if isinstance(fx.f.f.mean, ConstMean) else [])
    lp = logpdf(fx, y)
    grad = {}

    def step_for(v0):
        return rel_step * abs(v0) if v0 != 0.0 else rel_step

    for name, owner, attr in entries:
        if owner is None:
            v0 = float(fx.sigma2[0])
            hstep = step_for(v0)
            vals = []
            try:
                for sgn in (1.0, -1.0):
                    fx.sigma2 = np.array([v0 + sgn * hstep])
                    vals.append(logpdf(fx, y))
            finally:
                fx.sigma2 = np.array([v0])
        else:
            v0 = getattr(owner, attr)
            hstep = step_for(v0)
            vals = []
            try:
                for sgn in (1.0, -1.0):
                    setattr(owner, attr, v0 + sgn * hstep)
                    vals.append(logpdf(fx, y))
            finally:
                setattr(owner, attr, v0)
        grad[name] = (vals[0] - vals[1]) / (2 * hstep)
    return lp, grad


def logpdf_and_gradient(fx, y, rel_step=None, method=None):
    """(logpdf(fx, y), {name: d logpdf / d parameter}) for the kernel hyper-parameters (`parameters`), the noise
    variance ("noise") and a ConstMean ("mean.c"). The T-step work -- value and tangents -- runs on the device as
    forward-mode tangent scans or as one adjoint pass; the derivative of the O(1) host map parameter -> (A, Q, H, ..., x0) is EXACT
    (Van Loan block exponential for dA, dQ = dP - dA P A' - A dP A' - A P dA', class-by-class rules for F, H, P: `_components_jet`).
    `rel_step` only concerns the "fd" method (differences of the logpdf itself, default 1e-4) and the device-side differencing of
    exp(F dt_k) on irregular inputs.
    Regular spacing with homoscedastic noise: any supported state dimension. Irregular spacing: d <= 4, shared or per-step
    noise; the per-step tangents of exp(F dt_k) are formed on the device (tgp_logpdf_grad_sde).
    method: None (default policy), "adjoint" (ONE reverse-time pass on the stationary-gain engine, exact block tangents on the host:
    regular spacing, homoscedastic noise, d <= 8 -- cost independent of the number of parameters), "tangent" (the forward-mode scans)
    or "fd" (central differences of the device logpdf).
    Default: the adjoint pass where it applies, else tangent scans up to state dimension 8; from d = 9 (e.g. ApproxPeriodicKernel, d = 14) the dual-number kernels are
    out-of-line private-memory code (d = 14, T = 2e5: 2.4 s for 4 parameters against 5.7 ms per logpdf), so central differences
    of the logpdf -- 9 evaluations on the group kernels, ~50 ms, relative accuracy ~1e-7 -- are used instead.
    Heteroscedastic noise (one variance per observation) on the "fd" route: the kernel's and the mean's parameters only -- there is no
    single noise parameter to differentiate, the result has no "noise" entry."""
    if method not in (None, "tangent", "fd", "adjoint"):
        raise ValueError("method must be None, 'adjoint', 'tangent' or 'fd'")
    if method == "adjoint" or (method is None and isinstance(fx.x, RegularSpacing) and fx.sigma2.shape[0] == 1
                               and isinstance(fx.f.f.mean, (ZeroMean, ConstMean)) and fx.build_lgssm().dim <= 8):
        try:
            return _logpdf_and_gradient_adjoint(fx, y)
        except (L._lib.Unsupported, NotImplementedError):
            if method == "adjoint":
                raise                       # (default policy: the covariance did not settle within the head etc. -> the tangent scans below)
    if method == "fd" or (method is None and isinstance(fx.x, RegularSpacing) and fx.build_lgssm().dim >= 9):
        return _logpdf_and_gradient_fd(fx, y, rel_step=1e-4 if rel_step is None else rel_step)
    rel_step = 1e-6 if rel_step is None else rel_step
    if not isinstance(fx.x, RegularSpacing):
        # any plain array of inputs -- uniformly spaced or not -- is the reference's AbstractVector path (lti_sde.jl:135-146:
        # per-step blocks, dt_1 := 1), so its gradient goes through the SDE-described model; only RegularSpacing is LTI
        return _logpdf_and_gradient_sde(fx, y, rel_step)
    _shared_blocks(fx)                      # raises for layouts the gradient pass does not cover
    plist = parameters(fx.f.f.kernel)
    names = [n for n, _, _ in plist] + ["noise"] + (["mean.c"] if isinstance(fx.f.f.mean, ConstMean) else [])
    tangents = _shared_block_tangents(fx, names, plist)
    lp, g = L.logpdf_and_grad(fx.build_lgssm(), y, tangents)
    return lp, dict(zip(names, g))
