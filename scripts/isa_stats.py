"""Instruction statistics of kernels in a hipcc -S listing:  python scripts/isa_stats.py file.s [name-substring ...]"""
import re
import sys

txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
for m in re.finditer(r'^(_Z\w+):\s*; @\1\n(.*?)\n\s*s_endpgm', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pats and not any(p in name for p in pats):
        continue
    ins = [l.split()[0] for l in body.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    cnt = lambda p: sum(1 for i in ins if re.match(p, i))
    waitvm = len(re.findall(r's_waitcnt[^\n]*vmcnt', body))
    print(f"{name[:70]:70s} n={len(ins):6d} valu={cnt('v_'):6d} fma64={cnt('v_fma_f64|v_mul_f64|v_add_f64'):5d} salu={cnt('s_'):5d} "
          f"gload={cnt('global_load'):4d} gstore={cnt('global_store'):4d} sload={cnt('s_load'):4d} scratch={cnt('scratch_'):4d} "
          f"ds={cnt('ds_'):4d} bperm={cnt('ds_bpermute'):4d} dpp={sum(1 for l in body.split(chr(10)) if 'dpp' in l):4d} "
          f"waitvm={waitvm:4d} barrier={cnt('s_barrier'):3d} branch={cnt('s_cbranch'):4d} "
          f"readlane={cnt('v_readlane|v_writelane'):4d} accvgpr={cnt('v_accvgpr'):4d}")
