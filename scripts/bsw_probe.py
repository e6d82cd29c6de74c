#!/usr/bin/env python3
"""BSW-only throughput probe: python scripts/bsw_probe.py [Mpairs] [read_len] [distinct]
(distinct = 1: every pair its own extension job, as bench.py's bsw leg; default: 8 192 pairs tiled)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))
import numpy as np, torch
from pymeme import hipapi, workload
n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 2000000
rl = int(sys.argv[2]) if len(sys.argv) > 2 else 150
if len(sys.argv) > 3 and sys.argv[3] == "1":
    pairs, ref, qer = workload.make_bsw_pairs_distinct(n, seed=77, read_len=rl)
else:
    pairs, ref, qer, base = workload.make_bsw_pairs(n, seed=77, read_len=rl)
cells = int(np.minimum(pairs["len1"].astype(np.int64), 10**9).dot(np.ones(n, np.int64)))   # rows in total; cells are counted by the oracle in bench.py
ctx = hipapi.Context(0)
d_pairs = torch.from_numpy(pairs.view(np.uint8)).cuda(); d_ref = torch.from_numpy(ref).cuda(); d_qer = torch.from_numpy(qer).cuda()
torch.cuda.synchronize()
for it in range(4):
    ctx.bsw_batch_device(d_pairs.data_ptr(), d_ref.data_ptr(), d_qer.data_ptr(), n, 100)
    ctx.sync()
    print("[bsw probe] %s %d pairs (read_len %d): %.2f ms -> %.1f M pairs/s" % (os.path.basename(os.environ.get("MEME_HIP_LIB", "default")), n, rl, ctx.timings().bsw_kernel_ms, n / ctx.timings().bsw_kernel_ms / 1e3), flush=True)
