"""d = 14 (ApproxPeriodicKernel's default state size): time of every interface operation at T = 2e5, LTI and per-step layouts,
Forward and Reverse -- which ones still run on the out-of-line private-memory kernels."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import temporalgps_jl_amd as tgp
from tests import _util as U
from tests.test_gpu_parity import to_device_model

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 14
def timed(f, n=3):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n
CASES = [(c[0] == 'p', c[1]) for c in (sys.argv[3].split(',') if len(sys.argv) > 3 else ['lF', 'lR', 'pF', 'pR'])]
for tv, ordering in CASES:
    if True:
        rng = np.random.default_rng(1)
        model = U.random_lgssm(rng, tv, d, T if tv else T, ordering)
        dm = to_device_model(tgp, model)
        y = rng.standard_normal(T)
        eps = (rng.standard_normal((T, d)), rng.standard_normal(T), rng.standard_normal(d))
        res = {"logpdf": timed(lambda: tgp.logpdf(dm, y)), "filter": timed(lambda: tgp._filter(dm, y), 1),
               "marginals": timed(lambda: tgp.marginals(dm)), "rand": timed(lambda: tgp.rand(eps, dm))}
        if ordering == "F":
            res["posterior_marginals"] = timed(lambda: tgp.posterior_marginals(dm, y, np.array([0.1])))
        hd = dm.handle()
        hd.set_option(tgp._lib.OPT_PROFILE, 1); hd.profile_reset()
        tgp.logpdf(dm, y); tgp.marginals(dm); tgp.rand(eps, dm)
        names = sorted(k for k in hd.profile() if k.startswith(("k_reduce", "k_apply", "k_group_reduce", "k_group_apply", "k_group_affine", "k_group_marg")))
        print(f"d={d} {'per-step' if tv else 'lti'} {ordering}: " + ", ".join(f"{k} {v * 1e3:.1f} ms" for k, v in res.items()) + f" | {names}", flush=True)
