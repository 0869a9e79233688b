import sys
import numpy as np
sys.path.insert(0, ".")
import torch
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde, _lib
from oracle import components as oc, seq_kalman as sk
T = 50_000
rng = np.random.default_rng(1)
y_np = rng.standard_normal(T)
kern = lti_sde.Matern52Kernel() + 0.5 * lti_sde.Matern52Kernel().stretch(0.3)
spec = ("sum", ("matern52",), ("scaled", 0.5, ("stretched", 0.3, ("matern52",))))
ref = oc.build_lgssm(spec, ("regular", 0.0, 0.1, T), 0.1)
lp_ref = sk.logpdf(ref, y_np)
pm, pv = sk.posterior_marginals(ref, y_np, np.array([0.05]))
model = lti_sde.build_lgssm(kern, lti_sde.RegularSpacing(0.0, 0.1, T), 0.1)
hd = model.handle()
y = torch.as_tensor(y_np, device="cuda:0")
for variant in (1, 2):
    for fuse in (0, 1):
        hd.set_option(_lib.OPT_VARIANT, variant); hd.set_option(_lib.OPT_FUSE_SCAN, fuse)
        lp = tgp.logpdf(model, y)
        m, v = tgp.posterior_marginals(model, y, np.array([0.05]))
        print(f"RESULT variant={variant} fuse={fuse} lp_err={abs(lp-lp_ref)/abs(lp_ref):.2e} mean_err={np.max(np.abs(m.cpu().numpy()-pm)):.2e} var_err={np.max(np.abs(v.cpu().numpy()-pv)):.2e} kv={hd.lib.tgp_kernel_variant(hd.h)}")
