"""Development prototype of the stationary-gain engine's ONE-LAUNCH path (csrc/tgp_modal.hip), in NumPy with the kernel's structure:
the host plan comes from the product's own tgp_steady_plan (a pure host function of libtgp_hip.so), the rest restates k_steady_one --
sequential head in modal coordinates, workgroups of NW tiles of 64 lanes x 8 steps with `halo` steps of run-in / run-out from zero
states, in-tile Hillis-Steele scans with the element-wise powers M^(8 2^k), tile carries over the three neighbouring tiles, the
WJ / WG corrections.  tests/test_steady_plan.py holds it against the oracle on the CPU tier; the HIP kernel itself is checked in
tests/test_gpu_modal.py."""
import ctypes

import numpy as np

KS, TILE = 8, 512


def plan(model, T):
    import temporalgps_jl_amd as tgp
    lib = tgp._lib.load()
    d = len(model["x0m"])

    def blk(name, n):
        return np.ascontiguousarray(np.asarray(model[name], dtype=np.float64).reshape(-1)[:n])
    A = np.asarray(model["A"], dtype=np.float64).reshape(-1, d, d)[0]
    Q = np.asarray(model["Q"], dtype=np.float64).reshape(-1, d, d)[0]
    Ac, Qc = np.ascontiguousarray(A.T).reshape(-1), np.ascontiguousarray(Q.T).reshape(-1)      # column-major
    a, H, hh, R = blk("a", d), blk("H", d), blk("h", 1), blk("R", 1)
    x0m = np.ascontiguousarray(np.asarray(model["x0m"], dtype=np.float64))
    x0P = np.ascontiguousarray(np.asarray(model["x0P"], dtype=np.float64).reshape(d, d).T).reshape(-1)
    info_i, info_d = np.zeros(8, dtype=np.int32), np.zeros(4)
    modal, tabs = np.zeros(270), np.zeros(624 * (d * d + 2 * d + 3) + 2048 + 80)
    p = lambda x: x.ctypes.data
    rc = lib.tgp_steady_plan(d, p(Ac), p(a), p(Qc), p(H), p(hh), p(R), p(x0m), p(x0P), ctypes.c_int64(T), p(info_i), p(info_d), p(modal), p(tabs))
    assert rc == 0, rc
    out = dict(why=int(info_i[0]), n0=int(info_i[1]), n1=int(info_i[2]), nhs=int(info_i[3]), halo=int(info_i[4]), npair=int(info_i[5]),
               nw=int(info_i[6]), cond_f=info_d[0], cond_g=info_d[1], rho=info_d[2], resid=info_d[3], d=d)
    if out["why"] != 0:
        return out
    names = ["fd", "fo", "fb", "fa", "fw", "gd", "go", "gc", "gw", "fp8r", "fp8i", "gp8r", "gp8i", "fp512r", "fp512i", "gp512r", "gp512i"]
    for k, nm in enumerate(names):
        out[nm] = modal[8 * k:8 * k + d].copy()
    out["WJ"] = modal[136:200].reshape(8, 8)[:, :d].copy()
    out["WG"] = modal[200:264].reshape(8, 8)[:, :d].copy()
    out["hh"], out["rS"], out["iS"], out["logS"], out["LS"], out["vb"] = modal[264:270]
    n, dd = out["n0"] + 1, d * d
    q = 0
    out["h"] = tabs[0:d].copy()
    out["mu0"] = tabs[8:8 + d].copy()
    out["Wm"] = tabs[16:16 + dd].reshape(d, d).copy()
    q = 80
    for nm, cnt, shape in [("t_db", n * d, (n, d)), ("t_iS", n, (n,)), ("t_rS", n, (n,)), ("t_G", n * dd, (n, d, d)), ("t_c", n * d, (n, d)),
                           ("t_vb", n, (n,)), ("tvb", out["n1"], (out["n1"],))]:
        out[nm] = tabs[q:q + cnt].reshape(shape).copy()
        q += cnt
    return out


def _partner(d):
    return np.array([(i ^ 1) if (i ^ 1) < d else i for i in range(d)])


def _bmul(pr, pi, x, P):
    """x[..., d] <- block form (pr, pi) times x"""
    return pr * x + pi * x[..., P]


def _powers(re, im, nlev):
    pr, pi = [re.copy()], [im.copy()]
    for _ in range(1, nlev):
        a, b = pr[-1], pi[-1]
        pr.append(a * a - b * b)
        pi.append(2 * a * b)
    return pr, pi


def run(model, y, Rnew, post=True):
    """returns None when the plan says the path does not apply; else (lml, mean, var, plan)"""
    T = len(y)
    pl = plan(model, T)
    if pl["why"] != 0:
        return None
    d, n0, n1, nhs, halo, nw = pl["d"], pl["n0"], pl["n1"], pl["nhs"], pl["halo"], pl["nw"]
    P = _partner(d)
    fd, fo, fb, fa, fw = pl["fd"], pl["fo"], pl["fb"], pl["fa"], pl["fw"]
    gd, go, gc, gw = pl["gd"], pl["go"], pl["gc"], pl["gw"]
    fpr, fpi = _powers(pl["fp8r"], pl["fp8i"], 6)
    gpr, gpi = _powers(pl["gp8r"], pl["gp8i"], 6)
    ftr, fti = _powers(pl["fp512r"], pl["fp512i"], 2)
    gtr, gti = _powers(pl["gp512r"], pl["gp512i"], 2)
    hh, rS = pl["hh"], pl["rS"]
    Rn = np.broadcast_to(np.asarray(Rnew, dtype=np.float64).reshape(-1), (T,)) if np.size(Rnew) == 1 else np.asarray(Rnew, dtype=np.float64)
    mean, var = np.full(T, np.nan), np.full(T, np.nan)
    # ---- head forward (modal coordinates)
    z = pl["mu0"].copy()
    rh = np.zeros(nhs)
    quad = 0.0
    for t in range(nhs):
        ti = min(t, n0)
        u = y[t] - hh
        r = u - fw @ z
        quad += r * r * pl["t_iS"][ti]
        z = fd * z + fo * z[P] + fb * u + fa + pl["t_db"][ti] * r
        rh[t] = r
    z0 = z
    # ---- workgroups
    C = nw * TILE - 2 * halo
    nwg = (T - nhs + C - 1) // C
    ssq = 0.0
    lam_head = None
    for g in range(nwg):
        c_lo = nhs + g * C
        c_hi_raw = c_lo + C
        c_hi = min(c_hi_raw, T)
        s0 = nhs if g == 0 else c_lo - halo
        idx = s0 + np.arange(nw * TILE).reshape(nw, 64, KS)
        valid = idx < T
        yy = np.where(valid, y[np.minimum(idx, T - 1)], 0.0)
        any_valid = idx[:, 0, 0] < T
        # forward, zero start
        zz = np.zeros((nw, 64, d))
        r0 = np.zeros((nw, 64, KS))
        for j in range(KS):
            u = yy[:, :, j] - hh
            r0[:, :, j] = u - zz @ fw
            zz = fd * zz + fo * zz[..., P] + fb * u[..., None] + fa
        for k in range(6):
            off = 1 << k
            sh = np.zeros_like(zz)
            sh[:, off:] = zz[:, :-off]
            zz = zz + np.where((np.arange(64) >= off)[None, :, None], _bmul(fpr[k], fpi[k], sh, P), 0.0)
        zz[~any_valid] = 0.0
        F = zz[:, 63].copy()
        st = np.zeros_like(zz)
        st[:, 1:] = zz[:, :-1]
        zin = np.zeros((nw, d))
        for w in range(nw):
            for k in (1, 2, 3):
                src = w - k
                if src < -1 or (src == -1 and g != 0):
                    continue
                x = F[src] if src >= 0 else z0
                zin[w] += x if k == 1 else _bmul(ftr[k - 2], fti[k - 2], x, P)
        x = np.broadcast_to(zin[:, None, :], (nw, 64, d)).copy()
        lanes = np.arange(64)
        for k in range(6):
            px = _bmul(fpr[k], fpi[k], x, P)
            x = np.where(((lanes >> k) & 1).astype(bool)[None, :, None], px, x)
        st = st + x
        r = r0 - np.einsum("jd,wld->wlj", pl["WJ"][:, :d], st)
        r = np.where(valid, r, 0.0)
        in_core = (idx[:, :, 0] >= c_lo) & (idx[:, :, 0] < c_hi_raw)
        ssq += float(np.sum((r * r).sum(axis=2) * in_core))
        if not post:
            continue
        # backward, zero lam behind each tile
        ze = np.zeros((nw, 64, d))
        m0 = np.zeros((nw, 64, KS))
        for j in range(KS - 1, -1, -1):
            m0[:, :, j] = yy[:, :, j] - rS * r[:, :, j] + ze @ gw
            ze = gd * ze + go * ze[..., P] + gc * r[:, :, j][..., None]
        for k in range(6):
            off = 1 << k
            sh = np.zeros_like(ze)
            sh[:, :-off] = ze[:, off:]
            ze = ze + np.where((np.arange(64) + off < 64)[None, :, None], _bmul(gpr[k], gpi[k], sh, P), 0.0)
        need_back = any_valid & (idx[:, 0, 0] + TILE > c_lo)
        ze[~need_back] = 0.0
        B0 = ze[:, 0].copy()
        zst = np.zeros_like(ze)
        zst[:, :-1] = ze[:, 1:]

        def right_input(w):
            zi = np.zeros(d)
            for k in (1, 2, 3):
                src = w + k
                if src >= nw:
                    continue
                zi += B0[src] if k == 1 else _bmul(gtr[k - 2], gti[k - 2], B0[src], P)
            return zi
        zin = np.stack([right_input(w) for w in range(nw)])
        x = np.broadcast_to(zin[:, None, :], (nw, 64, d)).copy()
        back = 63 - lanes
        for k in range(6):
            px = _bmul(gpr[k], gpi[k], x, P)
            x = np.where(((back >> k) & 1).astype(bool)[None, :, None], px, x)
        zst = zst + x
        m = m0 + np.einsum("jd,wld->wlj", pl["WG"][:, :d], zst)
        sel = valid & (idx >= c_lo) & (idx < c_hi)
        mean[idx[sel]] = m[sel]
        tt = idx[sel]
        back_t = T - 1 - tt
        vbt = np.where(back_t < n1, pl["tvb"][np.minimum(back_t, n1 - 1)], pl["vb"])
        var[tt] = vbt + Rn[tt]
        if g == 0:
            lam_head = pl["Wm"] @ right_input(-1)
    # ---- head backward (original coordinates)
    if post:
        lam = lam_head
        for t in range(nhs - 1, -1, -1):
            ti = min(t, n0)
            mean[t] = y[t] - pl["t_rS"][ti] * rh[t] + pl["h"] @ lam
            lam = pl["t_G"][ti] @ lam + pl["t_c"][ti] * rh[t]
            var[t] = pl["t_vb"][ti] + Rn[t]
    lml = -0.5 * (T * np.log(2 * np.pi) + pl["LS"] + (T - n0) * pl["logS"] + quad + pl["iS"] * ssq)
    return lml, mean, var, pl
