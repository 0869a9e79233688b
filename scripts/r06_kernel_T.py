"""Round 6: hipEvent durations of the two headline kernels over series lengths: python scripts/r06_kernel_T.py [d]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from temporalgps_jl_amd import _lib as L  # noqa: E402
from temporalgps_jl_amd import lti_sde as P  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 3
kern = {3: ("matern52",), 2: ("matern32",), 1: ("matern12",)}[d]
for T in [int(float(a)) for a in sys.argv[2:]] or (10_000, 100_000, 1_000_000, 3_000_000, 10_000_000):
    model = P.build_lgssm(P.to_kernel(kern), P.RegularSpacing(0.0, 0.1, T), 0.1)
    hd = model.handle()
    y = torch.randn((T,), dtype=torch.float64, device="cuda:0")
    Rnew = torch.full((1,), 1e-18, dtype=torch.float64, device="cuda:0")
    mean, var = torch.empty_like(y), torch.empty_like(y)
    out = ctypes.c_double()
    yp, rp, mp, vp = L.ptr(y), L.ptr(Rnew), L.ptr(mean), L.ptr(var)
    for _ in range(5):
        hd.lib.tgp_logpdf(hd.h, yp, None, L.IN_DEVICE, ctypes.byref(out))
        hd.lib.tgp_posterior_marginals(hd.h, yp, None, rp, L.IN_DEVICE | L.OUT_DEVICE | L.SHARED_R, mp, vp, None)
    hd.set_option(L.OPT_PROFILE, 1)
    hd.profile_reset()
    for _ in range(30):
        hd.lib.tgp_logpdf(hd.h, yp, None, L.IN_DEVICE, ctypes.byref(out))
        hd.lib.tgp_posterior_marginals(hd.h, yp, None, rp, L.IN_DEVICE | L.OUT_DEVICE | L.SHARED_R, mp, vp, None)
    hd.set_option(L.OPT_PROFILE, 0)
    print(f"d {d} T {T:>9}: " + ", ".join(f"{k} {v['total_ms'] / v['calls'] * 1e3:.1f} us" for k, v in hd.profile().items()), flush=True)
