import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"].get("kernel"), round(d["roofline"].get("frac"),3), {k:round(v['avg_ms'],4) for k,v in d.get("kernels").items()})
print({k:(round(v.get("ms_per_step"),4) if isinstance(v,dict) else v) for k,v in d.items() if k.startswith("with_")})
