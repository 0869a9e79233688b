#!/usr/bin/env python3
"""Randomised check of the stationary-gain engine's ONE-LAUNCH path (TGP_OPT_STEADY = 3, csrc/tgp_modal.hip + tgp_steady_plan.hpp) against the
sequential C oracle: random kernels (sums of scaled, stretched Matern terms, d = 1..8), spacings 0.003..1, noise 1e-4..3, series lengths around
every tile / workgroup / head boundary, shared or per-step new noise, transition and emission offsets, host or device arrays (device pointers
on and off the 16-byte boundary).  Reports which engine served each case (the plan's verdict for the ones it declined) and holds the
five-launch engine (TGP_OPT_STEADY = 2) against the same reference.  usage: stress_modal.py [n_cases] [seed]"""
import ctypes
import gc
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import temporalgps_jl_amd as tgp  # noqa: E402
from oracle import components as oc  # noqa: E402
from oracle import seq_kalman as sk  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
START = int(os.environ.get("STRESS_START", "0"))      # first case to run (the earlier ones only advance the generator)
VERBOSE = os.environ.get("STRESS_VERBOSE", "") != ""      # print a case's parameters BEFORE running it
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
NAMES = ["matern12", "matern32", "matern52"]
DIM = dict(matern12=1, matern32=2, matern52=3)
LENGTHS = [300, 513, 600, 1023, 1024, 1025, 3583, 3584, 3585, 4095, 4096, 4097, 4608, 5000, 7552, 8192, 8193, 12345, 40960, 65536 + 511, 100_003, 250_000]
WHY = ["applies", "covariance not settled", "not positive definite", "series too short", "ill-conditioned modal form", "mixes too slowly", "tail too long", "eigenvalues"]
bad, one_launch, dense_one, draws = 0, 0, 0, 0
for case in range(n_cases):
    while True:
        terms = [(NAMES[rng.integers(3)], float(np.exp(rng.normal(0, 0.7))), float(np.exp(rng.normal(0, 0.7)))) for _ in range(rng.integers(1, 4))]
        if rng.random() < 0.35:      # a second summand with the SAME length scale (another variance): a defective closed loop -- no modal form,
            k = int(rng.integers(len(terms)))      # both recursions on dense powers (k_smooth_one)
            terms.append((terms[k][0], float(np.exp(rng.normal(0, 0.7))), terms[k][2]))
        if sum(DIM[t[0]] for t in terms) <= 8:
            break
    dt = float(np.exp(rng.uniform(np.log(0.003), np.log(1.0))))
    noise = float(np.exp(rng.uniform(np.log(1e-4), np.log(3.0))))
    T = int(LENGTHS[rng.integers(len(LENGTHS))])
    spec = tuple(("scaled", s2, ("stretched", s, (nm,))) for nm, s2, s in terms)
    spec = spec[0] if len(spec) == 1 else ("sum",) + spec
    model = oc.build_lgssm(spec, ("regular", 0.0, dt, T), noise)
    d = len(model["x0m"])
    if rng.random() < 0.3:          # offsets
        model["a"] = np.broadcast_to(0.05 * rng.standard_normal(d), np.asarray(model["a"]).shape).copy()
        model["h"] = np.broadcast_to(np.array(rng.standard_normal()), np.asarray(model["h"]).shape).copy()
    per_step_h = rng.random() < 0.2
    if per_step_h:                  # an emission offset PER STEP (a mean function at the inputs): the gains do not see it -- the one-launch structure holds
        model["h"] = np.sin(0.37 * np.arange(T)) * float(rng.standard_normal()) + 0.01 * np.arange(T) * float(rng.standard_normal())
    eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    Rn = np.exp(rng.normal(-2, 1, size=T)) if rng.random() < 0.3 else np.array([float(np.exp(rng.normal(-2, 1)))])
    mode = int(rng.integers(3))          # 0 host arrays, 1 device arrays, 2 device arrays 8 bytes past a 16-byte boundary
    if case < START:      # (replaying a sweep up to a case: the generator's draws of the cases skipped, none of their device work)
        if d <= 6 and T <= 70000:
            rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d)
        continue
    if VERBOSE:
        print(f"[{case:3d}] .. d={d} T={T} dt={dt:.4f} noise={noise:.2e} Rn={'T' if Rn.shape[0] > 1 else '1'} mode={mode} terms={terms}", flush=True)
    y = sk.rand(model, *eps)
    lp_ref = sk.logpdf(model, y)
    pm, pv = sk.posterior_marginals(model, y, Rn)
    msgs = []
    served = {}
    for opt in (3, 2):
        tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
        dm = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=T)
        dm.handle_options[tgp._lib.OPT_STEADY] = opt
        hd = dm.handle()
        hd.set_option(tgp._lib.OPT_PROFILE, 1)
        try:
            if mode == 0 or opt == 2:
                lp, mean, var = tgp.logpdf_and_posterior_marginals(dm, y, Rn)
            else:
                import torch
                off = 1 if mode == 2 else 0
                yb = torch.zeros(T + 1, dtype=torch.float64, device="cuda")
                yb[off:off + T] = torch.from_numpy(y).cuda()
                om, ov = torch.zeros(T + 1, dtype=torch.float64, device="cuda"), torch.zeros(T + 1, dtype=torch.float64, device="cuda")
                lp, mean, var = tgp.logpdf_and_posterior_marginals(dm, yb[off:off + T], torch.from_numpy(Rn).cuda(), out=(om[off:off + T], ov[off:off + T]))
                mean, var = mean.cpu().numpy(), var.cpu().numpy()
            names = set(hd.profile())
            served[opt] = ("one-launch" if any(n.startswith(("k_steady_one", "k_post_stream", "k_lml_stream")) for n in names) else "dense one-launch" if any(n.startswith("k_smooth_one") for n in names) and len(names) <= 2
                           else ("five-launch" if any(n.startswith("k_steady_apply") for n in names) else "general"))
            scale = max(1.0, float(np.max(np.abs(pm))))
            if not abs(lp - lp_ref) <= 1e-10 * abs(lp_ref):
                msgs.append(f"[{opt}] logpdf {lp} vs {lp_ref}")
            if not np.max(np.abs(mean - pm)) <= 1e-8 * scale:
                msgs.append(f"[{opt}] mean err {np.max(np.abs(mean - pm)):.2e}")
            if not np.max(np.abs(var - pv)) <= 1e-8 * max(1.0, float(np.max(pv))):
                msgs.append(f"[{opt}] var err {np.max(np.abs(var - pv)):.2e}")
            lp1 = tgp.logpdf(dm, y)
            if not abs(lp1 - lp_ref) <= 1e-10 * abs(lp_ref):
                msgs.append(f"[{opt}] logpdf-only {lp1} vs {lp_ref}")
            if opt == 3 and d <= 6 and T <= 70000:
                # a draw from the posterior: the one-launch path (tgp_posterior_rand) against the evaluated route (tgp_posterior, then
                # tgp_rand on the Reverse model -- the general engine), the same draws
                e2 = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
                hd.profile_reset()
                got = tgp.rand(e2, tgp.replace_observation_noise_cov(tgp.posterior(dm, y), Rn))
                drew = "k_smooth_one<rand>" in set(hd.profile())
                dm_e = tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=T)
                pe = tgp.replace_observation_noise_cov(tgp.posterior(dm_e, y), Rn)
                pe.materialise()
                want = tgp.rand(e2, pe)
                if not np.max(np.abs(got - want)) <= 1e-8 * max(1.0, float(np.max(np.abs(want)))):
                    msgs.append(f"posterior draw err {np.max(np.abs(got - want)):.2e} (one launch: {drew})")
                draws += drew
                del dm_e, pe
        except Exception as ex:      # noqa: BLE001
            msgs.append(f"[{opt}] {type(ex).__name__}: {ex}")
        del dm
        gc.collect()
    one_launch += served.get(3) == "one-launch"
    dense_one += served.get(3) == "dense one-launch"
    # the plan's own verdict for the record
    A = np.asarray(model["A"]).reshape(-1, d, d)[0]
    Q = np.asarray(model["Q"]).reshape(-1, d, d)[0]
    ii, dd = np.zeros(8, dtype=np.int32), np.zeros(4)
    p = lambda x: np.ascontiguousarray(x, dtype=np.float64).ctypes.data
    keep = [np.ascontiguousarray(A.T).reshape(-1), np.asarray(model["a"]).reshape(-1)[:d].copy(), np.ascontiguousarray(Q.T).reshape(-1), np.asarray(model["H"]).reshape(-1)[:d].copy(),
            np.asarray(model["h"]).reshape(-1)[:1].copy(), np.asarray(model["R"]).reshape(-1)[:1].copy(), np.asarray(model["x0m"], dtype=np.float64).copy(),
            np.ascontiguousarray(np.asarray(model["x0P"]).reshape(d, d).T).reshape(-1)]
    tgp._lib.load().tgp_steady_plan(d, *[k.ctypes.data for k in keep], ctypes.c_int64(T), ii.ctypes.data, dd.ctypes.data, None, None)
    if per_step_h:
        if served.get(3) != "dense one-launch" and ii[0] == 0 and T >= 1000:
            msgs.append(f"per-step emission offset served by {served.get(3)}")
    elif (ii[0] == 0) != (served.get(3) == "one-launch"):
        msgs.append(f"plan verdict {WHY[ii[0]]} but served by {served.get(3)}")
    bad += bool(msgs)
    print(f"[{case:3d}] {'FAIL' if msgs else 'ok'} d={d} T={T} dt={dt:.4f} noise={noise:.2e} Rn={'T' if Rn.shape[0] > 1 else '1'} mode={mode} terms={[(t[0][6:], round(t[1], 2), round(t[2], 2)) for t in terms]} "
          f"served={served.get(3)}/{served.get(2)} plan={WHY[ii[0]]} n0={ii[1]} halo={ii[4]} {'; '.join(msgs)}", flush=True)
print(f"{bad} failing cases of {n_cases}  ({one_launch} served by the one-launch path, {dense_one} by the dense-powers one-launch path; {draws} posterior draws in one launch)")
