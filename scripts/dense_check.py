"""Dense large-state path (d > 16): quick parity + timing check against the NumPy oracle (run on the GPU box)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib
from oracle import lgssm_ref as ref


def random_model(rng, T, d, p, ordering="F", per_step=False):
    def spd(n, scale=1.0):
        X = rng.standard_normal((n, n)) / np.sqrt(n)
        return scale * (X @ X.T + 0.5 * np.eye(n))
    nA = T if per_step else 1
    A = np.stack([0.9 * np.linalg.qr(rng.standard_normal((d, d)))[0] * rng.uniform(0.5, 1.0) for _ in range(nA)])
    a = rng.standard_normal((nA, d)) * 0.1
    Q = np.stack([spd(d, 0.3) for _ in range(nA)])
    H = rng.standard_normal((nA, p, d)) / np.sqrt(d)
    h = rng.standard_normal((nA, p)) * 0.1
    Rd = rng.uniform(0.05, 0.3, size=(T, p))
    R = np.stack([np.diag(r) for r in Rd])
    return dict(ordering=ordering, kind="small", T=T, A=A, a=a, Q=Q, H=H, h=h, R=R, x0m=rng.standard_normal(d), x0P=spd(d)), Rd


def to_dev(model, Rd):
    order = tgp.Forward if model["ordering"] == "F" else tgp.Reverse
    return tgp.LGSSM(tgp.GaussMarkovModel(order, model["A"], model["a"], model["Q"], tgp.Gaussian(model["x0m"], model["x0P"])),
                     tgp.SmallOutputLGC(model["H"], model["h"], Rd), T=model["T"])


def main():
    rng = np.random.default_rng(0)
    ok = True
    for (T, d, p, ordering, per_step, miss) in [(5, 20, 5, "F", False, False), (6, 48, 16, "F", True, True), (5, 40, 33, "R", False, False),
                                               (4, 192, 64, "F", False, True), (3, 300, 100, "R", True, False)]:
        model, Rd = random_model(rng, T, d, p, ordering, per_step)
        y = rng.standard_normal((T, p))
        dm = to_dev(model, Rd)
        if miss:
            mk = rng.random((T, p)) < 0.2
            lp_ref = ref.logpdf_missing(model, [np.where(mk[t], np.nan, y[t]) for t in range(T)], None) if False else None
            # oracle for per-element missing: the reference's rule (R_ii = 1e15, y_i = 0, + compensation)
            m2 = dict(model)
            R2 = model["R"].copy()
            y2 = y.copy()
            for t in range(T):
                for i in range(p):
                    if mk[t, i]:
                        R2[t][i, i] = 1e15
                        y2[t, i] = 0.0
            m2["R"] = R2
            lp_ref = ref.logpdf(m2, y2) + mk.sum() * 0.5 * np.log(2 * np.pi * 1e15)
            fm_ref, fP_ref = ref.filter_(m2, y2)
            yin = np.where(mk, np.nan, y)
        else:
            lp_ref = ref.logpdf(model, y)
            fm_ref, fP_ref = ref.filter_(model, y)
            yin = y
        lp = tgp.logpdf(dm, yin)
        fm, fP = tgp._filter(dm, yin)
        e1 = abs(lp - lp_ref) / abs(lp_ref)
        e2 = np.max(np.abs(fm - fm_ref)) / max(1.0, np.max(np.abs(fm_ref)))
        e3 = np.max(np.abs(fP - fP_ref)) / max(1.0, np.max(np.abs(fP_ref)))
        good = e1 < 1e-10 and e2 < 1e-9 and e3 < 1e-9
        ok &= good
        print(f"T={T} d={d} p={p} {ordering} per_step={per_step} miss={miss}: lml rel {e1:.2e}  m {e2:.2e}  P {e3:.2e}  {'ok' if good else 'FAIL'}", flush=True)
    # space-time at the BASELINE size (short T)
    from temporalgps_jl_amd import lti_sde, space_time
    from oracle import components as oc
    Nr, T = 256, 6
    r = np.linspace(-3, 3, Nr)
    k = space_time.Separable(space_time.SEKernel(), lti_sde.Matern52Kernel())
    grid = space_time.RectilinearGrid(r, lti_sde.RegularSpacing(0.0, 0.01, T))
    dmod = space_time.build_lgssm(k, grid, 0.1)
    model = oc.build_lgssm_separable(("se",), ("matern52",), r, ("regular", 0.0, 0.01, T), 0.1)
    Y = rng.standard_normal((T, Nr))
    t0 = time.time()
    lp_ref = ref.logpdf(model, Y)
    t1 = time.time()
    lp = tgp.logpdf(dmod, Y)
    print(f"space-time d=768 p=256 T={T}: lml {lp:.10f} ref {lp_ref:.10f} rel {abs(lp - lp_ref) / abs(lp_ref):.2e} (oracle {t1 - t0:.2f} s)", flush=True)
    ok &= abs(lp - lp_ref) < 1e-9 * abs(lp_ref)
    # the same with the reference's dense products (structure exploitation off)
    dm2 = space_time.build_lgssm(k, grid, 0.1)
    dm2.handle_options[_lib.OPT_DENSE_STRUCTURE] = 0
    lp2 = tgp.logpdf(dm2, Y)
    print(f"  dense products: lml {lp2:.10f} rel {abs(lp2 - lp_ref) / abs(lp_ref):.2e}; kernel variants {dmod.handle().lib.tgp_kernel_variant(dmod.handle().h)} / {dm2.handle().lib.tgp_kernel_variant(dm2.handle().h)}", flush=True)
    ok &= abs(lp2 - lp_ref) < 1e-9 * abs(lp_ref)
    # timing
    T = 400
    grid = space_time.RectilinearGrid(r, lti_sde.RegularSpacing(0.0, 0.01, T))
    dmod = space_time.build_lgssm(k, grid, 0.1)
    Y = rng.standard_normal((T, Nr))
    hd = dmod.handle()
    tgp.logpdf(dmod, Y)
    t0 = time.time()
    lp = tgp.logpdf(dmod, Y)
    dt = time.time() - t0
    print(f"timing: T={T} logpdf {dt * 1e3:.1f} ms -> {dt / T * 1e6:.1f} us/step, {2.57e9 * T / dt / 1e12:.2f} TF/s algorithmic", flush=True)
    print("kernel variant", hd.lib.tgp_kernel_variant(hd.h))
    hd.set_option(_lib.OPT_PROFILE, 1)
    hd.profile_reset()
    tgp.logpdf(dmod, Y)
    for name, v in hd.profile().items():
        print(f"  {name:28s} {v['total_ms'] / max(1, v['calls']) * 1e3:9.2f} us x {v['calls']}")
    dm3 = space_time.build_lgssm(k, grid, 0.1)
    dm3.handle_options[_lib.OPT_DENSE_STRUCTURE] = 0
    tgp.logpdf(dm3, Y)
    t0 = time.time()
    tgp.logpdf(dm3, Y)
    dt = time.time() - t0
    print(f"timing (dense products): T={T} logpdf {dt * 1e3:.1f} ms -> {dt / T * 1e6:.1f} us/step, {2.57e9 * T / dt / 1e12:.2f} TF/s algorithmic", flush=True)
    print("ALL OK" if ok else "SOME FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
