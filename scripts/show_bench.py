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
ir = d.get('irregular_spacing')
if ir:
    for lab in ("closed_form", "tiled_record"):
        print("irregular %-12s %.3e steps/s  %.3f ms/step" % (lab, ir[lab]['steps_per_s'], ir[lab]['ms_per_step']))
        for k, v in ir[lab]['kernels'].items():
            print(f"  {k:40s} {v['avg_ms']*1e3:8.1f} us")
    print("  lml rel diff %.1e" % ir['lml_rel_diff'])
