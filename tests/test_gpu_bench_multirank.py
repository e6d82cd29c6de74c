"""bench.py's multi-rank arrangement (one process per GPU: rank 0 builds the index on its GPU, the raw images are broadcast, every rank
stages its own copy and seeds its own reads, barrier + max-over-ranks timing) exercised on whatever the box has.  `python bench.py --gpus 2`
spawns its own ranks; on a box with one GPU the two ranks share it and talk over gloo (RCCL refuses two ranks on one device; on a multi-GPU
node the same command runs over RCCL, one rank per GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_2_self_started_one_json_line():
    env = dict(os.environ, MEME_BENCH_MBP="64", MEME_BENCH_READS="200000", MEME_BENCH_CACHE="0", MEME_BENCH_E2E_PAIRS="20000",
               MEME_BENCH_CPU_READS="20000", MEME_BENCH_BSW_PAIRS="100000")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True, env=env,
                       timeout=1500, cwd=REPO)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                      # rank 0 prints the one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["reads_per_gpu_per_step"] == 200000 and d["config"]["sample_parity_with_oracle"] is True
    assert d["config"]["reads_per_rank"] == [400000, 400000] and d["config"]["collective"]["ranks_seen"] == 2
    assert d["config"]["collective"]["index_broadcast_bytes"] > 5 * 2 * 64e6
    # what makes a real 8-GPU run interpretable: the broadcast's time and rate, every rank's staging and kernel times, the start-up budget
    col = d["config"]["collective"]
    assert col["index_broadcast_s"] > 0 and col["index_broadcast_GBps"] > 0 and len(col["per_rank"]) == 2
    assert all(r["staging_s"] > 0 and r["search_stage_ms"] > 0 and r["startup_s"] < 1800 for r in col["per_rank"]) and "start-up" in col["startup_budget"]
    # whole-job aggregate: both ranks' reads over the slower rank's time
    assert abs(d["value"] - 2 * 200000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # the N>1 line keeps the reported extras
    assert d["cpu_baseline"] and d["cpu_baseline"]["value"] > 0
    assert d["bsw"] and d["chain"] and d["chain"]["reads_left_to_host"] == 0
    if "skipped" not in d["e2e"]:
        assert d["e2e"]["sam_identical"] is True and d["e2e"]["gpus_driven_by_the_one_aligner_process"] == 2
