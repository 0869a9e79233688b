"""CPU tier: `python bench.py --gpus N ...` exactly as the driver invokes it (no torch.distributed environment) re-launches
itself under torch.distributed.run, shards ONE series over the ranks and prints one JSON line from rank 0. The per-segment
device work is replaced by the host emulation engine (tests/_bench_fake.py) and the backend is gloo: this box has no GPU."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--T", "300",
           "--engine-factory", "tests._bench_fake:make_engine", *extra]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


def test_bench_gpus2_self_launches_and_defaults_to_config4_strong_scaling():
    out = _run(2)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "strong"                       # N > 1 default: one series, total T fixed
    assert out["config"]["workload"] == "sum52_12_d4"       # BASELINE config 4's d = 4 kernel
    assert out["config"]["ranks"] == 2 and out["config"]["backend"] == "gloo"
    one = _run(1, ("--workload", "sum52_12_d4"))
    assert one["n_gpus"] == 1
    # same series, same log marginal likelihood whether it is sharded or not
    assert abs(out["logpdf"] - one["logpdf"]) <= 1e-10 * abs(one["logpdf"])


def test_bench_weak_scaling_is_an_option():
    out = _run(2, ("--scaling", "weak", "--workload", "matern32_d2"))
    assert out["scaling"] == "weak" and out["config"]["workload"] == "matern32_d2"
    assert np.isfinite(out["value"]) and out["value"] > 0
