"""bench.py under the driver's multi-GPU launcher, on the one GPU a test box has: a FRESH child
`python -m torch.distributed.run --nproc-per-node 1 bench.py ...` (the child has not touched the GPU before torchrun
starts it), so the nccl (= RCCL) initialisation, the stdout guard that keeps the JSON line alone on stdout
(distributed.py) and the hardware-queue setting (five streams per rank) are exercised without an 8-GPU node.
Reference aggregation: code/modelv2/modelseval.py:221-224 (means of per-image values)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["bench.py", "--steps", "10", "--warmup", "4", "--no-cpu-baseline"]


def _run(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines            # exactly one JSON line on stdout
    return json.loads(lines[0])


def test_bench_under_torchrun_on_one_gpu():
    _run([sys.executable] + ARGS)            # discarded: the first process on a fresh box pages the libraries in
    plain = _run([sys.executable] + ARGS)
    launched = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                     "--master-addr", "127.0.0.1", "--master-port", "29571"] + ARGS + ["--gpus", "1"])
    for d in (plain, launched):
        assert d["n_gpus"] == 1 and d["steps"] == 10 and d["unit"] == "images/s" and d["scaling"] == "weak"
        assert d["roofline"]["frac"] <= 1.0 and d["coder"]["ms_per_batch"] > 0
        assert abs(d["mean_bpp"] - plain["mean_bpp"]) < 1e-9
    ratio = launched["value"] / plain["value"]
    print(f"plain {plain['value']:.0f} images/s, under torch.distributed.run {launched['value']:.0f} ({ratio:.3f})")
    assert 0.88 < ratio < 1.14, (plain["value"], launched["value"])
