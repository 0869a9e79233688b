"""Probe: does torch.cuda initialise after libtgp_hip.so has already used the HIP runtime in this process?"""
import sys
import numpy as np
order = sys.argv[1]
sys.path.insert(0, ".")
if order == "torch_first_init":
    import torch
    torch.cuda.init()
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import lti_sde
if order == "torch_import_only":
    import torch
m = lti_sde.build_lgssm(lti_sde.Matern32Kernel(), lti_sde.RegularSpacing(0.0, 0.1, 1000), 0.1)
print("logpdf", tgp.logpdf(m, np.zeros(1000)))
import torch
try:
    torch.cuda.init()
    print(order, "torch.cuda.init OK", torch.cuda.device_count())
    y = torch.zeros(1000, dtype=torch.float64, device="cuda:0")
    print("logpdf dev", tgp.logpdf(m, y))
except Exception as e:
    print(order, "torch.cuda.init FAILED:", e)
