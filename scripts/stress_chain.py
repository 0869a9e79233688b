#!/usr/bin/env python3
"""ONE process, many minutes: the randomised sweeps one after the other in the same interpreter (the round-4 verdict's item 7: model objects,
handles, pinned buffers and streams of thousands of cases come and go without the HSA runtime running out of queues or scratch).
usage: stress_chain.py [scale] [seed offset]   (scale 1 ~ 6.5 minutes on one MI355X)"""
import os
import runpy
import sys
import time

here = os.path.dirname(os.path.abspath(__file__))
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # added to every sweep's seed
PLAN = [("stress_modal.py", 1200, 101), ("stress_steady.py", 700, 102), ("stress_sde.py", 120, 103), ("stress_gp_api.py", 60, 104), ("stress_modal.py", 800, 105),
        ("stress_general.py", 60, 106), ("stress_gradient.py", 40, 107)]
t00 = time.time()
failed = []
for name, n, seed in PLAN:
    n = max(1, int(n * scale))
    seed += seed0
    sys.argv = [name, str(n), str(seed)]
    t0 = time.time()
    print(f"==== {name} {n} cases, seed {seed} (t = {t0 - t00:.0f} s)", flush=True)
    try:
        runpy.run_path(os.path.join(here, name), run_name="__main__")
    except SystemExit as ex:
        if ex.code not in (None, 0):
            failed.append((name, ex.code))
    print(f"==== {name}: {time.time() - t0:.0f} s", flush=True)
print(f"chain of {len(PLAN)} sweeps in one process: {time.time() - t00:.0f} s; non-zero exits: {failed}")
