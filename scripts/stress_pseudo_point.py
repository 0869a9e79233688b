#!/usr/bin/env python3
"""Randomised check of the pseudo-point (DTC / VFE) path (space_time/pseudo_point.jl): sums of scaled separable kernels, N = 1..12 space
points, M = 1..5 pseudo-points (state dimension M * d_t up to 15), T = 1..60 regular or irregular times, scalar noise, missing points --
dtc, elbo and approx_posterior_marginals against the dense sparse-GP formulas (oracle/dense_gp.py), the reference's own test at its
tolerances (test/space_time/pseudo_point.jl:92-111).   usage: stress_pseudo_point.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import temporalgps_jl_amd  # noqa: E402,F401
from oracle import components as oc  # noqa: E402
from oracle import dense_gp as dg  # noqa: E402
from temporalgps_jl_amd import lti_sde as S  # noqa: E402
from temporalgps_jl_amd import pseudo_point as pp  # noqa: E402
from temporalgps_jl_amd import space_time as ST  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def one(spec):
    name = spec[0]
    if name == "se":
        return ST.SEKernel()
    if name == "stretched":
        return one(spec[2]).stretch(spec[1])
    return {"matern12": S.Matern12Kernel, "matern32": S.Matern32Kernel, "matern52": S.Matern52Kernel}[name]()


def product_kernel(terms):
    out = None
    for s, a, b in terms:
        k = ST.Separable(one(a), one(b))
        kk = k if s == 1.0 else s * k
        out = kk if out is None else out + kk
    return out


DT = dict(matern12=1, matern32=2, matern52=3)
bad = 0
for case in range(int(os.environ.get("START", "0")), min(n_cases, int(os.environ.get("END", "1000000")))):
    rng = np.random.default_rng([seed, case])
    while True:
        terms, dt_sum = [], 0
        for _ in range(int(rng.integers(1, 3))):
            tn = ["matern12", "matern32", "matern52"][rng.integers(3)]
            tk = (tn,) if rng.random() < 0.5 else ("stretched", float(np.exp(rng.normal(0, 0.4))), (tn,))
            sk_ = ("se",) if rng.random() < 0.6 else (["matern32", "matern52"][rng.integers(2)],)
            terms.append((1.0 if rng.random() < 0.3 else float(np.exp(rng.normal(0, 0.4))), sk_, tk))
            dt_sum += DT[tn]
        M = int(rng.integers(1, 6))
        if M * dt_sum <= 15:
            break
    N = int(rng.integers(1, 13))
    T = int(rng.choice([1, 2, 7, 30, 60]))
    dt = float(np.exp(rng.uniform(np.log(0.05), np.log(0.5))))
    regular = rng.random() < 0.6
    t = ("regular", 0.1, dt, T) if regular else np.cumsum(rng.random(T) * 2 * dt + 0.05 * dt)
    tt = oc.times(t)
    # (pseudo-points kept apart: two of them 2e-3 apart make K_zz's condition number 1e8, and the reference's jitters -- 1e-12 on K_zz, 1e-10
    #  in the Large-output update that the device's exact algebra does not need -- then show at 1e-4 in the result: seed 2, case 103 of the first run)
    r = rng.standard_normal(N) * 1.2
    z = (np.linspace(-1.5, 1.5, M) if M > 1 else np.zeros(1)) + 0.15 * rng.standard_normal(M)
    s2 = float(np.exp(rng.uniform(np.log(0.02), np.log(0.5))))
    msgs = []
    try:
        k = product_kernel(terms)
        grid = ST.RectilinearGrid(r, S.RegularSpacing(0.1, dt, T) if regular else t)
        x, zz = dg.grid_points(r, tt), dg.grid_points(z, tt)
        y = rng.standard_normal(len(x[0]))
        miss = rng.random(len(y)) < 0.2 if (len(y) > 3 and rng.random() < 0.5) else np.zeros(len(y), dtype=bool)
        ym = y.copy()
        ym[miss] = np.nan
        keep = ~miss
        xk = (x[0][keep], x[1][keep])
        noise = np.full(int(keep.sum()), s2)
        # The reference projects every term of a sum onto the pseudo-points with that term's OWN space kernel; the dense DTC / VFE
        # formulas with the sum kernel are the same approximation only when the terms share the space kernel (as all of the reference's
        # test kernels do, test/space_time/pseudo_point.jl:34-50). Otherwise the comparison is with the oracle's restatement of the
        # reference's state-space construction.
        one_space = len({str(tm[1]) for tm in terms}) == 1
        if one_space:
            d_dense = dg.dtc_dense(terms, xk, zz, noise, y[keep])
        else:
            d_dense = oc.dtc_statespace(terms, z, r, t, s2, y, missing=miss if miss.any() else None)
        got = pp.dtc(k, grid, s2, ym, z)
        if os.environ.get("VERBOSE"):
            print("   dtc: product", got, "oracle state space", oc.dtc_statespace(terms, z, r, t, s2, y, missing=miss if miss.any() else None), "dense", d_dense)
            if not miss.any():
                print("   elbo: product", pp.elbo(k, grid, s2, y, z), "oracle state space", oc.elbo_statespace(terms, z, r, t, s2, y), "dense", dg.elbo_dense(terms, x, zz, noise, y))
        if not abs(got - d_dense) <= 3e-6 * max(1.0, abs(d_dense)):
            msgs.append(f"dtc {got} vs dense {d_dense}")
        if not miss.any():
            e_dense = dg.elbo_dense(terms, x, zz, noise, y) if one_space else oc.elbo_statespace(terms, z, r, t, s2, y)
            ge = pp.elbo(k, grid, s2, y, z)
            if not abs(ge - e_dense) <= 1e-6 * max(1.0, abs(e_dense)):
                msgs.append(f"elbo {ge} vs dense {e_dense}")
            if not one_space:
                raise StopIteration
            x_pr = rng.standard_normal(int(rng.integers(1, 8)))
            xs = dg.grid_points(x_pr, tt)
            pm, pv = dg.vfe_posterior_marginals(terms, x, zz, noise, y, xs)
            gm, gv = pp.approx_posterior_marginals(k, grid, s2, y, z, x_pr)
            if not np.allclose(gm.reshape(-1), pm, rtol=1e-5, atol=1e-5):
                msgs.append(f"posterior mean {np.max(np.abs(gm.reshape(-1) - pm)):.2e}")
            if not np.allclose(np.sqrt(gv.reshape(-1)), np.sqrt(pv), rtol=1e-5, atol=1e-5):
                msgs.append(f"posterior std {np.max(np.abs(np.sqrt(gv.reshape(-1)) - np.sqrt(pv))):.2e}")
    except StopIteration:
        pass
    # RegularInTime with a different number of points per time slice (regular_in_time.jl:8-89) == the grid with the absent points missing
    try:
        if regular and T * N > 2 and not msgs:
            keep2 = rng.random((T, N)) < 0.7
            if T > 2:
                keep2[T // 2] = False               # a slice with no observation at all
            keep2[0, 0] = True
            sig_full = 0.05 + 0.2 * rng.random((T, N)) if rng.random() < 0.5 else np.full((T, N), s2)
            y_full = rng.standard_normal((T, N))
            times = S.RegularSpacing(0.1, dt, T)
            ragged = ST.RegularInTime(times, [r[keep2[i]] for i in range(T)])
            g2 = ST.RectilinearGrid(r, times)
            ym2 = np.where(keep2, y_full, np.nan)
            a_, b_ = pp.dtc(k, ragged, sig_full[keep2], y_full[keep2], z), pp.dtc(k, g2, sig_full, ym2, z)
            if not abs(a_ - b_) <= 1e-8 * max(1.0, abs(b_)):
                msgs.append(f"ragged dtc {a_} vs grid with missing {b_}")
            a_, b_ = pp.elbo(k, ragged, sig_full[keep2], y_full[keep2], z), pp.elbo(k, g2, sig_full, ym2, z)
            if not abs(a_ - b_) <= 1e-8 * max(1.0, abs(b_)):
                msgs.append(f"ragged elbo {a_} vs grid with missing {b_}")
            x_pr = rng.standard_normal(3)
            m_r, v_r = pp.approx_posterior_marginals(k, ragged, sig_full[keep2], y_full[keep2], z, x_pr)
            m_g, v_g = pp.approx_posterior_marginals(k, g2, sig_full, ym2, z, x_pr)
            if not (np.allclose(m_r, m_g, rtol=0, atol=1e-8) and np.allclose(v_r, v_g, rtol=0, atol=1e-8)):
                msgs.append(f"ragged posterior {np.max(np.abs(m_r - m_g)):.2e} {np.max(np.abs(v_r - v_g)):.2e}")
    except Exception as ex:      # noqa: BLE001
        import traceback
        msgs.append(f"ragged: {type(ex).__name__}: {ex} @ {traceback.extract_tb(ex.__traceback__)[-1].lineno}")
    except Exception as ex:      # noqa: BLE001
        import traceback
        msgs.append(f"{type(ex).__name__}: {ex} @ {traceback.extract_tb(ex.__traceback__)[-1].lineno}")
    bad += bool(msgs)
    print(f"[{case:3d}] {'FAIL' if msgs else 'ok'} d={M * dt_sum} N={N} M={M} T={T} {'regular' if regular else 'irregular'} missing={int(miss.sum()) if 'miss' in dir() else '?'} terms={terms} {'; '.join(msgs)}", flush=True)
print(f"{bad} failing cases of {n_cases}")
