"""Per-step layout, T = 1e7: pass times of logpdf under each kernel-table variant (TGP_OPT_VARIANT 0 auto, 1 safe, 2 inlined)."""
import sys
sys.path.insert(0, ".")
import torch
import bench
import temporalgps_jl_amd as tgp
from temporalgps_jl_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else "sum52_52_d6"
T = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
model = bench.build_model(tgp, name, T, "per_step", 0)
hd = model.handle()
print("auto variant code", hd.lib.tgp_kernel_variant(hd.h))
y = torch.randn((T,), dtype=torch.float64, device="cuda:0")
for v in (0, 1, 2):
    hd.set_option(_lib.OPT_VARIANT, v)
    for _ in range(2):
        lp = tgp.logpdf(model, y)
    hd.set_option(_lib.OPT_PROFILE, 1)
    hd.profile_reset()
    for _ in range(3):
        lp = tgp.logpdf(model, y)
    torch.cuda.synchronize()
    prof = hd.profile()
    hd.set_option(_lib.OPT_PROFILE, 0)
    print("variant", v, "lp", lp, {k: round(s["total_ms"] / s["calls"] * 1e3, 1) for k, s in prof.items()})
