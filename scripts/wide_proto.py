"""The ALGORITHM of the wide-state engine (csrc/tgp_wide.hip) restated in NumPy -- the host plan (covariance to its fixed point, halo by
repeated squaring, the Bryson-Frazier backward matrices, the variance tables of the series' two ends) and the kernels' structure (chunks that warm up
`halo` steps early from zero, the observer row, the head on the host) -- so that the CPU tier can hold it against the oracle without a GPU
(tests/test_wide_proto.py).  Not the product path: the HIP kernels are checked by tests/test_gpu_wide.py."""
import numpy as np


def plan(model, T):
    A, a, Q = model["A"][0], model["a"][0], model["Q"][0]
    h, hh, R = model["H"][0], float(np.atleast_1d(model["h"])[0]), float(np.atleast_1d(model["R"])[0])
    d = len(a)
    P = 0.5 * (model["x0P"] + model["x0P"].T)
    Ks, Ss, Pfs = [], [], []
    prev, rate, prev_S = np.inf, 0.0, 0.0
    for t in range(8192):
        Pp = A @ P @ A.T + Q
        v = Pp @ h
        S = float(h @ v + R)
        Pf = Pp - np.outer(v, v) / S
        Pf = 0.5 * (Pf + Pf.T)
        chg, scale = np.max(np.abs(Pf - P)), np.max(np.abs(Pf))
        Ks.append(v / S)
        Ss.append(S)
        Pfs.append(Pf)
        P = Pf
        if chg > 1e-11 * scale and prev > 1e-11 * scale and chg < prev:
            rate = chg / prev
        still = chg <= 4 * 2.220446049250313e-16 * scale or (t >= 16 and chg >= prev and chg <= 1e-13 * scale)
        if chg == 0.0 or (still and chg <= (1 - rate) * 1e-12 * scale and abs(S - prev_S) <= (1 - rate) * 1e-13 * S):
            break
        prev, prev_S = chg, S
    else:
        return None
    n0 = len(Ks)
    K, S = Ks[-1], Ss[-1]
    g = A.T @ h
    Phi = A - np.outer(K, g)

    def halo_of(M):
        k, Mk = 1, M.copy()
        while np.abs(Mk).sum(axis=1).max() > 2.0 ** -60:
            Mk, k = Mk @ Mk, 2 * k
            if k > 2 ** 20:
                return None
        return k
    AK = A @ K
    Psi = A.T - np.outer(h, AK)
    gw = R * AK
    # the series' end: partial sums of (gw' Psi^k h)^2 / S
    q, u, qtab = 0.0, h.copy(), [0.0]
    for k in range(8192):
        c = float(gw @ u)
        q += c * c / S
        qtab.append(q)
        if (np.abs(gw).sum() * np.abs(u).max()) ** 2 / S <= 1e-20 * max(q, 1e-300) and k >= 2:
            break
        u = Psi @ u
    # Lam_inf by doubling, then the head's variances backwards
    Lam, M = np.outer(h, h) / S, Psi.copy()
    for _ in range(40):
        if np.abs(M).sum(axis=1).max() <= 1e-12:
            break
        Lam, M = Lam + M @ Lam @ M.T, M @ M
    headvar = np.zeros(n0)
    for t in range(n0 - 1, -1, -1):
        AKt = A @ Ks[t]
        gwt = R * AKt
        headvar[t] = (Ss[t] - R) * R / Ss[t] - gwt @ Lam @ gwt
        Pt = A.T - np.outer(h, AKt)
        Lam = np.outer(h, h) / Ss[t] + Pt @ Lam @ Pt.T
    return dict(d=d, n0=n0, Ks=np.array(Ks), Ss=np.array(Ss), K=K, S=S, Phi=Phi, g=g, g0=float(h @ a), c=a - K * float(h @ a), halo=halo_of(Phi), Psi=Psi, gw=gw,
                halo_b=halo_of(Psi), qtab=np.array(qtab), n1=len(qtab) - 1, vbase=(S - R) * R / S, headvar=headvar, A=A, a=a, h=h, hh=hh, R=R, x0m=model["x0m"],
                sum_logS_head=float(np.sum(np.log(Ss))))


def run(pl, y, Rnew, chunks=7):
    """logpdf and posterior marginals as the kernels compute them: the head on the host, `chunks` chunks behind it"""
    T, d, n0 = len(y), pl["d"], pl["n0"]
    A, a, h, hh, R = pl["A"], pl["a"], pl["h"], pl["hh"], pl["R"]
    # head forward
    m, quad, head_r = pl["x0m"].copy(), 0.0, np.zeros(n0)
    for t in range(n0):
        mp = A @ m + a
        r = y[t] - hh - h @ mp
        head_r[t] = r
        quad += r * r / pl["Ss"][t]
        m = mp + pl["Ks"][t] * r
    z0 = m
    Tb = T - n0
    ln = -(-Tb // chunks)
    r_all = np.zeros(T)
    ssq = 0.0
    for c in range(chunks):      # forward: lanes i < d hold z, the observer lane holds the row -g
        s0, s1 = n0 + c * ln, min(T, n0 + (c + 1) * ln)
        if s0 >= s1:
            continue
        from_head = s0 - pl["halo"] <= n0
        z = z0.copy() if from_head else np.zeros(d)
        for t in range(n0 if from_head else s0 - pl["halo"], s1):
            u = y[t] - hh
            r = u - pl["g"] @ z - pl["g0"]              # the observer's multiply-adds
            z = pl["Phi"] @ z + pl["K"] * u + pl["c"]   # the other lanes'
            if t >= s0:
                ssq += r * r
                r_all[t] = r
    lml = -0.5 * (T * np.log(2 * np.pi) + pl["sum_logS_head"] + (T - n0) * np.log(pl["S"]) + quad + ssq / pl["S"])
    mean, var = np.zeros(T), np.zeros(T)
    lam_n0 = None
    for c in range(chunks):      # backward: lam_t = h r_t / S + Psi lam_(t+1); the observer row gw gives mean_t - y_t + (R / S) r_t
        s0, s1 = n0 + c * ln, min(T, n0 + (c + 1) * ln)
        if s0 >= s1:
            continue
        lam = np.zeros(d)
        for t in range(min(T, s1 + pl["halo_b"]) - 1, s0 - 1, -1):
            if t < s1:
                mean[t] = y[t] - (R / pl["S"]) * r_all[t] + pl["gw"] @ lam
                jt = T - 1 - t
                var[t] = pl["vbase"] - (pl["qtab"][jt] if jt < pl["n1"] else pl["qtab"][-1]) + Rnew[t]
            lam = pl["h"] * r_all[t] / pl["S"] + pl["Psi"] @ lam
        if c == 0:
            lam_n0 = lam
    lam = lam_n0
    for t in range(n0 - 1, -1, -1):      # the head backwards on the host
        AKt = A @ pl["Ks"][t]
        akl = AKt @ lam
        mean[t] = y[t] - (R / pl["Ss"][t]) * head_r[t] + R * akl
        var[t] = pl["headvar"][t] + Rnew[t]
        lam = h * (head_r[t] / pl["Ss"][t] - akl) + A.T @ lam
    return lml, mean, var
