#!/usr/bin/env python3
"""Kernel-only throughput probe: python scripts/quick_probe.py [Mbp] [Mreads] [bits] [lanes,...]  (index cached in /dev/shm)"""
import os, sys, time, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from pymeme import hipapi, synth, workload

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 512
mreads = float(sys.argv[2]) if len(sys.argv) > 2 else 4
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lanes_list = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [4, 8]
log = lambda s: print("[probe]", s, flush=True)
g = synth.make_genome(int(mbp * 1e6), seed=11)
tmp = "/dev/shm/qprobe_%d_%d" % (int(mbp), bits)
if os.path.exists(tmp + "/done"):
    prefix = open(tmp + "/done").read().strip()
else:
    os.makedirs(tmp, exist_ok=True)
    prefix = workload.build_index_on_disk(g, tmp, bits=bits, log=log)
    open(tmp + "/done", "w").write(prefix)
lib = os.environ.get("MEME_HIP_LIB", "")
ctx = hipapi.Context(0)
ctx.load_index_files(prefix)
n = int(mreads * 1e6)
RL = int(os.environ.get("PROBE_READ_LEN", "150")); SUB = float(os.environ.get("PROBE_SUB_RATE", "0.01"))
reads = workload.make_reads_fast(g, n, RL, seed=12, sub_rate=SUB)
d_reads = torch.from_numpy(reads.reshape(-1)).cuda()
d_off = torch.arange(0, (n + 1) * RL, RL, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for lanes in lanes_list:
    ctx.set_tuning("group_lanes", lanes)
    for rounds in [int(x) for x in os.environ.get("PROBE_ROUNDS", "1,3").split(",")]:
      for defer in [int(x) for x in os.environ.get("PROBE_DEFER", "0,1").split(",")]:
        ctx.set_tuning("seed_defer", defer)
        for it in range(2):
            res = ctx.seed_batch_device(d_reads.data_ptr(), d_off.data_ptr(), n, n * RL, hipapi.default_seed_opt(rounds=rounds))
            tm = ctx.timings()
        log("%s len=%d sub=%.3f G=%d rounds=%d defer=%d: kernel %.1f ms (of which re-seeding kernel %.2f, %d lane searches) gather %.2f -> %.2f M reads/s; searches/read %.1f windows/read %.2f smems %d hits %d" % (
            os.path.basename(lib), RL, SUB, lanes, rounds, defer, tm.seed_kernel_ms, tm.seed_reseed_ms, tm.seed_lane_searches, tm.seed_gather_ms,
            n / tm.seed_kernel_ms / 1e3, res.searches / n, tm.seed_windows / n, res.total_smems, res.total_hits))
