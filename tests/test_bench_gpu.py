"""bench.py as the driver runs it: `python bench.py --gpus N` must start its own ranks (the reference spawns its ranks
itself, /root/reference/main/main.py:255-259).  Rehearsed on the one-GPU box with FRHIP_BENCH_BACKEND=gloo: both ranks on cuda:0, gloo
collectives on device tensors, the whole N > 1 code path (DataParallel arena all-reduce, class-sharded PartialFC at rate 0.1)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_starts_its_own_ranks_and_reports_them():
    env = dict(os.environ, FRHIP_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extra",
                        "--batch", "64", "--classes", "4000"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 128 and rec["scaling"] == "weak"
    assert rec["config"]["collective_backend"] == "gloo" and "rate 0.1" in rec["config"]["workload"]
    assert rec["value"] > 0 and rec["final_loss"] == rec["final_loss"]
