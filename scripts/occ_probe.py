#!/usr/bin/env python3
import os, sys, time, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from pymeme import hipapi, hostapi, synth, workload
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 1000000
g = synth.make_genome(int(mbp * 1e6), seed=11)
text, sa = hostapi.build_sa(g); l1, l2 = hostapi.train_prmi(text, sa)
ctx = hipapi.Context(0)
pp = np.zeros((sa.shape[0], 5), np.uint8); pp[:, :4] = (sa >> np.uint64(8)).astype('<u4').view(np.uint8).reshape(-1, 4); pp[:, 4] = (sa & np.uint64(255)).astype(np.uint8)
ctx.load_index_host(pp.reshape(-1), text, l1, l2)
reads = workload.make_reads_fast(g, n, 150, seed=12)
d_reads = torch.from_numpy(reads.reshape(-1)).cuda(); d_off = torch.arange(0, (n + 1) * 150, 150, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
bl = [int(x) for x in os.environ.get("BPC", "1,2,4").split(",")]
for lanes in [int(x) for x in os.environ.get("LANES", "8,16").split(",")]:
    for bpc in bl:
        ctx.set_tuning("group_lanes", lanes); ctx.set_tuning("seed_blocks_per_cu", bpc)
        for it in range(2):
            res = ctx.seed_batch_device(d_reads.data_ptr(), d_off.data_ptr(), n, n * 150, hipapi.default_seed_opt(rounds=3))
            tm = ctx.timings()
        print("G=%d blocks/CU=%d kernel %.1f ms -> %.2f M reads/s  windows/search %.2f" % (lanes, bpc, tm.seed_kernel_ms, n / tm.seed_kernel_ms / 1e3, tm.seed_windows / res.searches), flush=True)
