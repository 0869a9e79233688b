import json, sys
d = json.load(open(sys.argv[1]))
print("LTI   %.3e steps/s  %.3f ms/step  frac %.3f" % (d['value'], d['ms_per_step'], d['roofline']['frac']))
for k, v in d['kernels'].items():
    print(f"  {k:40s} {v['avg_ms']*1e3:8.1f} us")
g = d.get('roofline_general_layout')
if g:
    print("general %.3e steps/s  %.3f ms/step  frac %.3f" % (g['steps_per_s'], g['ms_per_step'], g['frac']))
    for k, v in g['kernels'].items():
        print(f"  {k:40s} {v['avg_ms']*1e3:8.1f} us")
