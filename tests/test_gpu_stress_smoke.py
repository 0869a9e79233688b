"""GPU tier: a seeded subset of every randomised sweep under scripts/stress_*.py (the full sweeps -- hundreds of cases each -- are run by hand
and recorded in DESIGN 2).  Each script draws random models / inputs, holds the device results against the oracle and ends with
"<n> failing cases of <N>"; the defects those sweeps found in earlier rounds (a kernel table chosen by the wrong family's verdict, a noise
diagonal read at the wrong row) passed the fixed-grid suite for two rounds, hence a slice of them in the driver's run."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWEEPS = [            # (script, cases, seed)
    ("stress_modal.py", 40, 11),
    ("stress_steady.py", 24, 12),
    ("stress_general.py", 30, 13),
    ("stress_general2.py", 14, 14),
    ("stress_sde.py", 16, 15),
    ("stress_gp_api.py", 24, 16),
    ("stress_multi.py", 16, 17),
    ("stress_gradient.py", 8, 18),
    ("stress_space_time.py", 12, 19),
    ("stress_pseudo_point.py", 16, 20),
    ("stress_long.py", 5, 21),      # series of 1e6 - 6e6 steps: thousands of workgroups, the sequential head, rand in one launch
    ("stress_wide.py", 5, 22),      # the wide-state engine (8 < d <= 63): random LTI models and products of kernels against the engines of before
]


@pytest.mark.parametrize("script,cases,seed", SWEEPS, ids=[s[0][:-3] for s in SWEEPS])
def test_seeded_slice_of_the_randomised_sweep(script, cases, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), str(cases), str(seed)], capture_output=True, text=True, timeout=900, cwd=ROOT)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-25:])
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) failing cases of (\d+)", r.stdout)
    assert m, tail
    assert int(m.group(1)) == 0 and int(m.group(2)) == cases, "\n".join(l for l in r.stdout.splitlines() if "FAIL" in l) + "\n" + tail
