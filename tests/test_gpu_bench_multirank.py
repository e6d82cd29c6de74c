"""bench.py's multi-rank arrangement (one process per GPU: rank 0 builds the index on its GPU, the raw images are broadcast, every rank
stages its own copy and seeds its own reads, barrier + max-over-ranks timing) exercised on whatever the box has: two ranks that share
GPU 0 and talk over gloo (RCCL refuses two ranks on one device; on a multi-GPU node the driver launches the same code over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_one_json_line():
    env = dict(os.environ, MEME_BENCH_DEVICE="0", MEME_BENCH_BACKEND="gloo", MEME_BENCH_MBP="64", MEME_BENCH_READS="200000", MEME_BENCH_CACHE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, env=env, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                      # rank 0 prints the one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["reads_per_gpu_per_step"] == 200000 and d["config"]["sample_parity_with_oracle"] is True
    # whole-job aggregate: both ranks' reads over the slower rank's time
    assert abs(d["value"] - 2 * 200000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
