"""bench.py's N > 1 control flow on a one-GPU box, every round: two ranks share device 0 (CS_BENCH_SHARE_GPU=1: the collectives of the
sharded BA go through the torch.distributed callback over gloo -- RCCL refuses two ranks on one device), launched exactly as the driver
launches the scaling run (python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2).  Exercises the rendezvous, the
per-rank frame shards, the barrier + max-over-ranks timing, the sharded BA leg behind its watchdog (cs_ba_set_shard, ownership, the
separator-mode trial with its three collectives) and the watchdog itself.  A functional check, never a performance number."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--steps", "4", "--warmup", "1", "--steady-steps", "0", "--latency-calls", "0", "--lines-images", "0", "--rp-frames", "0", "--no-edge", "--no-measure-traffic",
         "--no-cpu-baseline", "--frames", "200", "--unique", "20"]


def _run(port, extra_env):
    env = {**os.environ, "CS_BENCH_SHARE_GPU": "1", "MASTER_ADDR": "127.0.0.1", **extra_env}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2"] + QUICK
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                      # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_on_one_device_run_the_whole_bench_script():
    d = _run(29531, {})
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    ba = d["ba"]
    assert "error" not in ba, ba
    mg = ba["multi_gpu"]
    assert mg["ranks"] == 2 and mg["bytes_exchanged_per_trial"] > 0
    assert ba["value"] > 0 and ba["iterations"] >= 1


@pytest.mark.gpu
def test_two_ranks_watchdog_of_the_sharded_ba_leg_fires_and_the_line_survives():
    d = _run(29533, {"CS_BENCH_BA_LIMIT_S": "0.2"})
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert "error" in d["ba"], d["ba"]
