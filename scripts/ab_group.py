"""A/B on one box: logpdf of d = 5..8 LTI models through the group-per-chunk kernels (TGP_OPT_GROUP 1) against the
lane-per-chunk kernels (0); values against the sequential oracle at T = 2e5, timings at T = 1e7."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde, _lib
from oracle import components as oc, seq_kalman as sk
SPECS = {5: ("sum", ("matern52",), ("matern32",)), 6: ("sum", ("matern52",), ("matern52",)),
         7: ("sum", ("matern52",), ("matern32",), ("matern32",)), 8: ("sum", ("matern52",), ("matern52",), ("matern32",))}
for d, spec in SPECS.items():
    T = 200_000
    rng = np.random.default_rng(d)
    y = rng.standard_normal(T)
    miss = rng.random(T) < 0.1
    ref = oc.build_lgssm(spec, ("regular", 0.0, 0.1, T), 0.1)
    lp_ref = sk.logpdf(ref, y)
    model = lti_sde.build_lgssm(lti_sde.to_kernel(spec), lti_sde.RegularSpacing(0.0, 0.1, T), 0.1)
    hd = model.handle()
    yd = torch.as_tensor(y, device="cuda:0")
    res = {}
    for grp in (0, 2):
        hd.set_option(_lib.OPT_GROUP, grp)
        res[grp] = tgp.logpdf(model, yd)
    print(f"RESULT d={d} variant={hd.lib.tgp_kernel_variant(hd.h)} rel err lane {abs(res[0]-lp_ref)/abs(lp_ref):.2e} group {abs(res[2]-lp_ref)/abs(lp_ref):.2e}")
    T = 10_000_000
    model = lti_sde.build_lgssm(lti_sde.to_kernel(spec), lti_sde.RegularSpacing(0.0, 0.1, T), 0.1)
    hd = model.handle()
    yd = torch.randn(T, dtype=torch.float64, device="cuda:0")
    for grp in (0, 2):
        hd.set_option(_lib.OPT_GROUP, grp)
        for _ in range(3): tgp.logpdf(model, yd)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): tgp.logpdf(model, yd)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5 * 1e3
        hd.set_option(_lib.OPT_PROFILE, 1); hd.profile_reset()
        for _ in range(3): tgp.logpdf(model, yd)
        prof = hd.profile(); hd.set_option(_lib.OPT_PROFILE, 0)
        print(f"RESULT d={d} group={grp} logpdf {wall:.3f} ms | " + " ".join(f"{k.replace('k_','')}={v['total_ms']/v['calls']*1e3:.0f}" for k, v in prof.items()))
