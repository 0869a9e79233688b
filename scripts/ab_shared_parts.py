"""Pass 1 with the chunks' shared matrix parts from a table (TGP_OPT_SHARED_PARTS) against the general pass 1: same numbers, time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib, lti_sde as P

T = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
KERNELS = {"matern32 d=2": P.Matern32Kernel(), "matern52 d=3": P.Matern52Kernel(), "52+12 d=4": P.Matern52Kernel() + P.Matern12Kernel(),
           "52+32 d=5": P.Matern52Kernel() + P.Matern32Kernel(), "52+52 d=6": P.Matern52Kernel() + P.Matern52Kernel().stretch(0.7)}
for name, k in KERNELS.items():
    out = {}
    for opt in (0, 1):
        fx = P.to_sde(P.GP(k))(P.RegularSpacing(0.0, 0.1, T), 0.1)
        model = fx.build_lgssm()
        model.handle_options[_lib.OPT_SHARED_PARTS] = opt      # 1: default policy (table built on a side stream by the second call)
        y = torch.randn(T, dtype=torch.float64, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(1))
        Rn = torch.full((1,), 0.1, dtype=torch.float64, device="cuda:0")
        res = None
        for _ in range(3):
            lp = tgp.logpdf(model, y); res = tgp.logpdf_and_posterior_marginals(model, y, Rn)
            torch.cuda.synchronize(); time.sleep(0.02)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): tgp.logpdf(model, y)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(5): tgp.logpdf_and_posterior_marginals(model, y, Rn)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        hd = model.handle(); hd.set_option(_lib.OPT_PROFILE, 1); hd.profile_reset(); tgp.logpdf(model, y); prof = hd.profile(); hd.set_option(_lib.OPT_PROFILE, 0)
        p1 = [f"{k2} {v['total_ms'] / v['calls']:.3f}" for k2, v in prof.items() if k2.startswith(("k_reduce_filter", "k_filter_table"))]
        out[opt] = (lp, res[0], res[1].cpu().numpy(), res[2].cpu().numpy())
        print(f"{name} T={T} shared_parts={opt}: logpdf {(t1 - t0) / 5 * 1e3:.3f} ms, logpdf+posterior marginals {(t2 - t1) / 5 * 1e3:.3f} ms | {p1}", flush=True)
    a, b = out[0], out[1]
    print(f"   lml rel diff {abs(a[0] - b[0]) / abs(a[0]):.2e} / {abs(a[1] - b[1]) / abs(a[1]):.2e}, mean max diff {np.abs(a[2] - b[2]).max():.2e}, var max diff {np.abs(a[3] - b[3]).max():.2e}", flush=True)
