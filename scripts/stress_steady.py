#!/usr/bin/env python3
"""Randomised check of the stationary-gain engine against the sequential C oracle: random kernels (sums of scaled, stretched Matern terms),
spacings, noise levels and series lengths (around the tile / workgroup / head boundaries), single handle and 2-4 shards on one GPU,
logpdf + posterior marginals + adjoint-vs-tangent gradient. usage: stress_steady.py [n_cases] [seed]"""
import ctypes
import gc
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import temporalgps_jl_amd as tgp  # noqa: E402
from oracle import components as oc  # noqa: E402
from oracle import seq_kalman as sk  # noqa: E402
from temporalgps_jl_amd import lti_sde as P  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
NAMES = ["matern12", "matern32", "matern52"]
DIM = dict(matern12=1, matern32=2, matern52=3)
LENGTHS = [513, 514, 600, 1023, 1024, 1025, 4095, 4096, 4097, 4608, 4609, 5000, 8192, 8193, 12345, 40960, 65536 + 511, 100_003]
bad = 0
for case in range(n_cases):
    while True:
        terms = [(NAMES[rng.integers(3)], float(np.exp(rng.normal(0, 0.7))), float(np.exp(rng.normal(0, 0.7)))) for _ in range(rng.integers(1, 4))]
        if sum(DIM[t[0]] for t in terms) <= 8:
            break
    dt = float(np.exp(rng.uniform(np.log(0.003), np.log(1.0))))
    noise = float(np.exp(rng.uniform(np.log(1e-4), np.log(3.0))))
    T = int(LENGTHS[rng.integers(len(LENGTHS))])
    spec = tuple(("scaled", s2, ("stretched", s, (nm,))) for nm, s2, s in terms)
    spec = spec[0] if len(spec) == 1 else ("sum",) + spec
    model = oc.build_lgssm(spec, ("regular", 0.0, dt, T), noise)
    d = len(model["x0m"])
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    Rn = np.array([float(np.exp(rng.normal(-2, 1)))])
    W = int(rng.integers(2, 5))
    if case < int(os.environ.get("START", "0")):
        continue
    if os.environ.get("VERBOSE"):
        print(f"[{case:3d}] d={d} T={T} dt={dt} noise={noise} terms={terms} W={W}", flush=True)
    y = sk.rand(model, *eps)
    lp_ref = sk.logpdf(model, y)
    pm, pv = sk.posterior_marginals(model, y, Rn)
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=T)
    msgs = []
    lp, mean, var = tgp.logpdf_and_posterior_marginals(dm, y, Rn)
    if os.environ.get("VERBOSE"):
        print("   combined call done", flush=True)
    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    hd = dm.handle()
    hd.lib.tgp_steady_steps(hd.h, ctypes.byref(a), ctypes.byref(b))
    served = a.value > 0 and b.value - a.value <= 2048   # (a longer head is the general engine's mean-only count: the covariance had not settled in 2048 steps)
    if os.environ.get("VERBOSE"):
        print(f"   steps served with stationary gains: {a.value} of {b.value} (head: {b.value - a.value})", flush=True)
    scale = max(1.0, float(np.max(np.abs(pm))))
    if not abs(lp - lp_ref) <= 1e-10 * abs(lp_ref):
        msgs.append(f"logpdf {lp} vs {lp_ref}")
    if not np.max(np.abs(mean - pm)) <= 1e-8 * scale:
        msgs.append(f"mean err {np.max(np.abs(mean - pm)):.2e}")
    if not np.max(np.abs(var - pv)) <= 1e-8 * max(1.0, float(np.max(pv))):
        msgs.append(f"var err {np.max(np.abs(var - pv)):.2e}")
    lp1 = tgp.logpdf(dm, y)
    if not abs(lp1 - lp_ref) <= 1e-10 * abs(lp_ref):
        msgs.append(f"logpdf-only {lp1} vs {lp_ref}")
    if T >= 4 * W:
        ms = tgp.MultiLGSSM(dm, devices=[0] * W)
        lpm, mm, vm = ms.logpdf_and_posterior_marginals(y, Rn)
        if os.environ.get("VERBOSE"):
            print("   multi done", flush=True)
        if not abs(lpm - lp_ref) <= 1e-10 * abs(lp_ref):
            msgs.append(f"multi[{W}] logpdf {lpm} vs {lp_ref}")
        if not (np.max(np.abs(mm - pm)) <= 1e-8 * scale and np.max(np.abs(vm - pv)) <= 1e-8 * max(1.0, float(np.max(pv)))):
            msgs.append(f"multi[{W}] marginals {np.max(np.abs(mm - pm)):.2e} {np.max(np.abs(vm - pv)):.2e}")
    # gradient: adjoint against the tangent scans
    ks = [P.ScaledKernel(s2, P.StretchedKernel(s, P.to_kernel((nm,)))) for nm, s2, s in terms]
    k = ks[0]
    for kk in ks[1:]:
        k = k + kk
    fx = P.to_sde(P.GP(k))(P.RegularSpacing(0.0, dt, T), noise)
    try:
        lpa, ga = P.logpdf_and_gradient(fx, y, method="adjoint")
        if os.environ.get("VERBOSE"):
            print("   adjoint done", flush=True)
        if d <= 6:      # (the dual-number kernels of d = 7, 8 need a large scratch arena per HIP queue; the runtime aborts a queue when the arenas of
                        #  the queues alive in ONE process add up to ~300-480 MB, DESIGN 9 -- a sweep over many models in one process trips it)
            lpt, gt = P.logpdf_and_gradient(fx, y, method="tangent")
            sc = max(abs(v) for v in gt.values())
            err = max(abs(ga[n] - gt[n]) for n in gt) / sc
            if not err <= 1e-7:
                msgs.append(f"gradient adjoint vs tangent {err:.2e}")
        elif not abs(lpa - lp_ref) <= 1e-10 * abs(lp_ref):
            msgs.append(f"adjoint logpdf {lpa} vs {lp_ref}")
    except tgp._lib.Unsupported as refusal:
        # (the adjoint runs on the five-launch form of the engine: it may refuse what the one-launch form served -- a series shorter than that
        #  form's head and tail tiles -- but nothing the five-launch form itself serves)
        tr2 = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
        d2 = tgp.LGSSM(tr2, tgp.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=T)
        d2.handle_options[tgp._lib.OPT_STEADY] = 2
        h2 = d2.handle()
        h2.set_option(tgp._lib.OPT_PROFILE, 1)
        tgp.logpdf_and_posterior_marginals(d2, y, Rn)
        names = set(h2.profile())
        # (the engine's kernels are enqueued either way; where it did not apply the general engine's passes follow them)
        if any(n.startswith("k_steady_apply") for n in names) and not any(n.startswith(("k_reduce_filter", "k_apply_filter", "k_smooth", "k_group_", "k_sweep")) for n in names):
            msgs.append(f"adjoint refused a model the five-launch engine served: {refusal}")
        del d2
    del dm, fx
    gc.collect()          # (handles own HIP streams: the runtime's per-queue scratch arenas add up over the handles alive in a process)
    tag = "FAIL" if msgs else "ok"
    bad += bool(msgs)
    print(f"[{case:3d}] {tag} d={d} T={T} dt={dt:.4f} noise={noise:.2e} terms={[(t[0][6:], round(t[1], 2), round(t[2], 2)) for t in terms]} served={served} {'; '.join(msgs)}",
          flush=True)
print(f"{bad} failing cases of {n_cases}")
