"""Instruction mix of one kernel of tgp_modal.hip (device assembly): modal_isa.py <mangled-name-substring> [asm file].
Generate the assembly with: hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only tgp_modal.hip -o /tmp/tgp_modal.s"""
import collections
import re
import sys

want = sys.argv[1]
s = open(sys.argv[2] if len(sys.argv) > 2 else "/tmp/tgp_modal.s").read()
for m in re.finditer(r"\n(_ZN9tgp_modal\w+): *;[^\n]*\n", s):
    name = m.group(1)
    if want not in name:
        continue
    body = s[m.end():].split(".Lfunc_end")[0]
    ins = [l.strip().split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    g = lambda pred: sum(v for k, v in c.items() if pred(k))
    print(name)
    print(dict(total=len(ins), f64=g(lambda k: k.split("_e")[0] in ("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64")), dpp=g(lambda k: "dpp" in k),
               v_mov=g(lambda k: k.startswith("v_mov")), cndmask=g(lambda k: k.startswith("v_cndmask")), lanes=g(lambda k: "readlane" in k or "writelane" in k),
               scratch=g(lambda k: k.startswith("scratch")), glob=g(lambda k: k.startswith("global")), ds=g(lambda k: k.startswith("ds_")),
               waitcnt=c["s_waitcnt"], s_load=g(lambda k: k.startswith("s_load")), s_nop=c["s_nop"],
               salu=g(lambda k: k.startswith("s_") and not k.startswith(("s_waitcnt", "s_load", "s_nop"))), valu=g(lambda k: k.startswith("v_")),
               accvgpr=g(lambda k: "accvgpr" in k)))
    print(c.most_common(25))
