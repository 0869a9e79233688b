"""Development check of the stationary-gain scan engine (TGP_OPT_STEADY = 2) on the GPU box: parity against the sequential C oracle
for several kernels / lengths, which engine served the call, and a timing of the headline step.
    python scripts/steady2_check.py [--big]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import temporalgps_jl_amd as tgp          # noqa: E402
from oracle import components as oc      # noqa: E402
from oracle import seq_kalman as sk      # noqa: E402

OPT_STEADY, OPT_PROFILE = 12, 2


def device_model(model, T, steady):
    dev = tgp.LGSSM(
        tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"])),
        tgp.ScalarOutputLGC(model["H"], model["h"], model["R"]), T=T)
    dev.handle_options[OPT_STEADY] = steady
    return dev


def steady_steps(dev):
    import ctypes
    hd = dev.handle()
    a, b = ctypes.c_int64(), ctypes.c_int64()
    hd.check(hd.lib.tgp_steady_steps(hd.h, ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


def case(kern, dt, T, s2=0.1, mean=None, rnew=1e-18, per_step_rnew=False, seed=1):
    model = oc.build_lgssm(kern, ("regular", 0.0, dt, T), s2, mean)
    d = len(model["x0m"])
    rng = np.random.default_rng(seed)
    y = sk.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    lp_ref = sk.logpdf(model, y)
    Rn = np.full(T, rnew) * (1.0 + rng.random(T)) if per_step_rnew else np.array([rnew])
    m_ref, v_ref = sk.posterior_marginals(model, y, Rn)
    out = {}
    for steady in (2, 1):
        dev = device_model(model, T, steady)
        lp = tgp.logpdf(dev, y)
        s_lp = steady_steps(dev)
        mean_, var_ = tgp.posterior_marginals(dev, y, Rn)
        s_pm = steady_steps(dev)
        lp2, mean2, var2 = tgp.logpdf_and_posterior_marginals(dev, y, Rn)
        out[steady] = (abs(lp - lp_ref) / abs(lp_ref), abs(lp2 - lp_ref) / abs(lp_ref), np.max(np.abs(mean_ - m_ref)), np.max(np.abs(var_ - v_ref)),
                       np.max(np.abs(mean2 - m_ref)), s_lp[0], s_pm[0])
    ok = out[2][0] < 1e-10 and out[2][1] < 1e-10 and out[2][2] < 1e-8 and out[2][3] < 1e-8 and out[2][4] < 1e-8
    print(f"{'OK ' if ok else 'BAD'} {kern} dt={dt} T={T} d={d}: steady2 lml {out[2][0]:.1e}/{out[2][1]:.1e} mean {out[2][2]:.1e} var {out[2][3]:.1e} "
          f"[served {out[2][5]}/{out[2][6]} of {T}] | general lml {out[1][0]:.1e} mean {out[1][2]:.1e} var {out[1][3]:.1e}", flush=True)
    return ok


def timing(kern, dt, T, steps=20, steady=2):
    import torch
    model = oc.build_lgssm(kern, ("regular", 0.0, dt, T), 0.1)
    d = len(model["x0m"])
    rng = np.random.default_rng(2)
    y = sk.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
    dev = device_model(model, T, steady)
    yd = torch.from_numpy(y).cuda()
    Rn = torch.tensor([1e-18], dtype=torch.float64).cuda()
    outm = torch.empty(T, dtype=torch.float64, device="cuda")
    outv = torch.empty(T, dtype=torch.float64, device="cuda")
    for _ in range(5):
        tgp.logpdf_and_posterior_marginals(dev, yd, Rn, out=(outm, outv))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tgp.logpdf_and_posterior_marginals(dev, yd, Rn, out=(outm, outv))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    t0 = time.perf_counter()
    for _ in range(steps):
        tgp.logpdf(dev, yd)
    torch.cuda.synchronize()
    ms_lp = (time.perf_counter() - t0) / steps * 1e3
    dev.handle().set_option(OPT_PROFILE, 1)
    dev.handle().profile_reset()
    for _ in range(steps):
        tgp.logpdf_and_posterior_marginals(dev, yd, Rn, out=(outm, outv))
    prof = dev.handle().profile()
    dev.handle().set_option(OPT_PROFILE, 0)
    print(f"timing {kern} d={d} T={T} steady={steady}: combined {ms:.3f} ms ({T / ms / 1e3:.3e} steps/s), logpdf {ms_lp:.3f} ms; served {steady_steps(dev)}")
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
        print(f"    {k:45s} {v['total_ms'] / max(v['calls'], 1) * 1e3:9.1f} us x {v['calls'] // steps}")


if __name__ == "__main__":
    big = "--big" in sys.argv
    ok = True
    ok &= case(("matern52",), 0.1, 3000)
    ok &= case(("matern32",), 0.1, 2100)
    ok &= case(("matern12",), 0.1, 700)
    ok &= case(("matern52",), 0.1, 10_000, mean=("const", 1.5))
    ok &= case(("sum", ("matern52",), ("matern32",)), 0.1, 5000)
    ok &= case(("sum", ("matern52",), ("matern52",)), 0.1, 5000)
    ok &= case(("sum", ("matern52",), ("matern12",)), 0.1, 4099)
    ok &= case(("matern52",), 0.1, 4099)
    ok &= case(("matern52",), 0.03, 5000)
    ok &= case(("matern52",), 0.01, 20_000)
    ok &= case(("matern32",), 0.1, 10_000, per_step_rnew=True, rnew=0.1)
    ok &= case(("matern52",), 0.1, 600)          # shorter than head + tail: general path
    ok &= case(("sum", ("matern52",), ("matern32",), ("matern32",)), 0.1, 3000)
    ok &= case(("sum", ("matern52",), ("matern52",), ("matern32",)), 0.1, 3000)
    ok &= case(("scaled", 2.5, ("stretched", 0.7, ("matern32",))), 0.1, 2500, s2=0.5)
    ok &= case(("matern52",), 0.1, 1_000_003)
    print("ALL OK" if ok else "SOME BAD", flush=True)
    timing(("matern52",), 0.1, 10_000_000)
    timing(("matern52",), 0.1, 10_000_000, steady=1)
    timing(("matern32",), 0.1, 10_000_000)
    timing(("sum", ("matern52",), ("matern12",)), 0.1, 10_000_000)
    timing(("sum", ("matern52",), ("matern32",)), 0.1, 10_000_000)
    timing(("sum", ("matern52",), ("matern52",)), 0.1, 10_000_000)
    timing(("sum", ("matern52",), ("matern52",)), 0.1, 10_000_000, steady=1)
    timing(("matern52",), 0.1, 10_000)
    if big:
        timing(("sum", ("matern52",), ("matern12",)), 0.1, 100_000_000)
