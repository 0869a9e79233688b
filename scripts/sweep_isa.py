"""Instruction mix of the sweep engine's kernels (device assembly of tgp_sweep.hip): totals per kernel and per loop body."""
import collections
import re
import subprocess
import sys

src = "temporalgps.jl_amd/csrc/tgp_sweep.hip"
out = "/tmp/tgp_sweep.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-command-line-argument",
                       "-S", "--cuda-device-only", src, "-o", out])
s = open(out).read()
want = sys.argv[1:] or ["Li3ELb0ELb1E", "Li3ELb0ELb0E", "Li3ELb1ELb1E"]
parts = re.split(r"\n(_ZN9tgp_sweep7k_sweep\w+): *;[^\n]*\n", s)


def stats(ins):
    c = collections.Counter(ins)
    g = lambda pred: sum(v for k, v in c.items() if pred(k))
    return dict(total=len(ins), f64=g(lambda k: k.split("_e")[0] in ("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64")), accvgpr=g(lambda k: "accvgpr" in k),
                v_mov=g(lambda k: k.startswith("v_mov")), cndmask=c["v_cndmask_b32"], lanes=g(lambda k: "readlane" in k or "writelane" in k),
                scratch=g(lambda k: k.startswith("scratch")), glob=g(lambda k: k.startswith("global")), ds=g(lambda k: k.startswith("ds_")),
                waitcnt=c["s_waitcnt"], s_load=g(lambda k: k.startswith("s_load")), salu=g(lambda k: k.startswith("s_") and not k.startswith(("s_waitcnt", "s_load", "s_nop"))),
                valu=g(lambda k: k.startswith("v_")), rcp=c["v_rcp_f64_e32"] + c["v_rcp_f64_e64"], rsq=c["v_rsq_f64_e32"] + c["v_rsq_f64_e64"])


for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
    if not any(w in name for w in want):
        continue
    lines = [l for l in body.split("\n")]
    ins = [l.strip().split()[0] for l in lines if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    print(name, stats(ins))
    # basic blocks by label; report the large ones
    blocks, cur, label = [], [], "entry"
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((label, cur))
            cur, label = [], m.group(1)
        elif l.startswith("\t") and not l.strip().startswith((".", ";")):
            cur.append(l.strip().split()[0])
    blocks.append((label, cur))
    for lab, b in blocks:
        if len(b) >= 150:
            st = stats(b)
            print(f"   {lab:12s}", {k: v for k, v in st.items() if v})
