"""Round 6: the streaming logpdf kernel (csrc/tgp_lml.hip) against the oracle's sequential restatement (oracle/seq_kalman.c), d = 1..8, lengths
around every tile / run / workgroup boundary, device pointers off the 16-byte boundary; then its time at T = 1e7 beside k_steady_one's
(TGP_LML_STREAM=0 in a second process).  Usage: python scripts/r06_lml_check.py [parity|time]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import temporalgps_jl_amd as tgp  # noqa: E402
from oracle import components as oc  # noqa: E402
from oracle import seq_kalman as sk  # noqa: E402

KERNELS = {
    1: ("matern12",),
    2: ("matern32",),
    3: ("matern52",),
    4: ("sum", ("matern52",), ("matern12",)),
    5: ("sum", ("matern52",), ("matern32",)),
    6: ("sum", ("matern52",), ("stretched", 0.4, ("matern52",))),
    7: ("sum", ("matern52",), ("stretched", 0.5, ("matern32",)), ("scaled", 0.3, ("matern32",))),
    8: ("sum", ("matern52",), ("stretched", 2.0, ("matern52",)), ("stretched", 0.5, ("matern32",))),
}


def device_model(model):
    tr = tgp.GaussMarkovModel(tgp.Forward, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"]))
    return tgp.LGSSM(tr, tgp.ScalarOutputLGC(model["H"], np.atleast_1d(model["h"]), np.atleast_1d(model["R"])), T=model["T"])


def kernels_of(dm, fn):
    hd = dm.handle()
    hd.set_option(tgp._lib.OPT_PROFILE, 1)
    hd.profile_reset()
    out = fn()
    names = dict(hd.profile())
    hd.set_option(tgp._lib.OPT_PROFILE, 0)
    return out, names


def draw(model, seed):
    T, d = model["T"], len(model["x0m"])
    rng = np.random.default_rng(seed)
    return sk.rand(model, rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))


def parity():
    worst = 0.0
    bad = 0
    for d, kern in KERNELS.items():
        for T in (70, 100, 700, 2047 + 64, 2048 + 64, 2049 + 64, 4096 + 64, 5000, 10_000, 16_384 + 64 + 1, 65_536, 300_001, 2048 * 2048 + 64 + 17, 5_000_000 if d <= 4 else 1_000_003):
            for (dt, s2) in ((0.1, 0.1), (0.01, 1e-3)) if T < 400_000 else ((0.1, 0.1),):
                model = oc.build_lgssm(kern, ("regular", 0.0, dt, T), s2)
                y = draw(model, 7 * d + T % 13)
                ref = sk.logpdf(model, y)
                dm = device_model(model)
                lp, names = kernels_of(dm, lambda: tgp.logpdf(dm, y))
                err = abs(lp - ref) / abs(ref)
                worst = max(worst, err)
                ok = err <= 1e-10
                bad += not ok
                print(f"d {d} T {T:>8} dt {dt} s2 {s2}: rel err {err:.2e} {'ok' if ok else 'FAIL'} kernels {sorted(names)}", flush=True)
    # device-resident observations off the 16-byte boundary
    import torch
    for d in (3, 6):
        T = 300_001
        model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.1)
        y = draw(model, 5)
        buf = torch.zeros(T + 1, dtype=torch.float64, device="cuda")
        buf[1:] = torch.from_numpy(y).cuda()
        dm = device_model(model)
        lp = tgp.logpdf(dm, buf[1:])
        ref = sk.logpdf(model, y)
        err = abs(lp - ref) / abs(ref)
        worst = max(worst, err)
        bad += not err <= 1e-10
        print(f"d {d} T {T} unaligned device pointer: rel err {err:.2e}")
    print(f"worst {worst:.2e}, failures {bad}")
    return bad


def timing():
    import torch
    for d in (3, 1, 2, 4, 5, 6, 8):
        T = 10_000_000
        model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.1)
        y = draw(model, 1)
        yd = torch.from_numpy(y).cuda()
        dm = device_model(model)
        for _ in range(5):
            lp = tgp.logpdf(dm, yd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            lp = tgp.logpdf(dm, yd)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        _, names = kernels_of(dm, lambda: [tgp.logpdf(dm, yd) for _ in range(20)])
        print(f"d {d}: logpdf call {ms * 1e3:.1f} us; kernels { {k: round(v['total_ms'] / max(v['calls'], 1) * 1e3, 2) for k, v in names.items()} } lml {lp:.6f}", flush=True)


def post_parity():
    import torch
    worst_m = worst_v = worst_l = 0.0
    bad = 0
    for d in (3, 1, 2):
        kern = KERNELS[d]
        for T in (1100, 1024 + 64, 2048 + 64 - 1, 2048 + 64, 2048 + 64 + 1, 5000, 10_000, 16_384 + 64 + 3, 100_000, 300_001, 2048 * 1024 + 64, 2048 * 1024 + 64 + 1025, 3_000_017):
            for (dt, s2) in ((0.1, 0.1), (0.01, 1e-3)) if T < 400_000 else ((0.1, 0.1),):
                model = oc.build_lgssm(kern, ("regular", 0.0, dt, T), s2)
                y = draw(model, 3 * d + T % 11)
                rng = np.random.default_rng(T)
                for per_step in (False, True):
                    Rn = rng.random(T) + 0.05 if per_step else np.array([0.3])
                    m_ref, v_ref = sk.posterior_marginals(model, y, Rn)
                    lp_ref = sk.logpdf(model, y)
                    dm = device_model(model)
                    (lp, mean, var), names = kernels_of(dm, lambda: tgp.logpdf_and_posterior_marginals(dm, y, Rn))
                    em, ev, el = np.max(np.abs(mean - m_ref)), np.max(np.abs(var - v_ref)), abs(lp - lp_ref) / abs(lp_ref)
                    worst_m, worst_v, worst_l = max(worst_m, em), max(worst_v, ev), max(worst_l, el)
                    ok = em <= 1e-8 and ev <= 1e-8 and el <= 1e-10
                    bad += not ok
                    print(f"d {d} T {T:>8} dt {dt} s2 {s2} per-step {per_step}: mean {em:.2e} var {ev:.2e} lml {el:.2e} {'ok' if ok else 'FAIL'} {sorted(names)}", flush=True)
                    if not ok:
                        i = int(np.argmax(np.abs(mean - m_ref)))
                        print("   worst mean at", i, "of", T, "; first bad:", int(np.argmax(np.abs(mean - m_ref) > 1e-8)))
    print(f"worst mean {worst_m:.2e} var {worst_v:.2e} lml {worst_l:.2e}, failures {bad}")
    return bad


def post_timing():
    import torch
    for d in (3, 1, 2):
        T = 10_000_000
        model = oc.build_lgssm(KERNELS[d], ("regular", 0.0, 0.1, T), 0.1)
        y = draw(model, 1)
        yd = torch.from_numpy(y).cuda()
        Rn = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda")
        out = (torch.empty_like(yd), torch.empty_like(yd))
        dm = device_model(model)
        for _ in range(5):
            tgp.posterior_marginals(dm, yd, Rn, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            tgp.posterior_marginals(dm, yd, Rn, out=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        _, names = kernels_of(dm, lambda: [tgp.posterior_marginals(dm, yd, Rn, out=out) for _ in range(20)])
        print(f"d {d}: posterior call {ms * 1e3:.1f} us; kernels { {k: round(v['total_ms'] / max(v['calls'], 1) * 1e3, 2) for k, v in names.items()} }", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "parity"
    if what == "parity":
        sys.exit(1 if parity() else 0)
    if what == "post":
        sys.exit(1 if post_parity() else 0)
    if what == "posttime":
        post_timing()
    else:
        timing()
