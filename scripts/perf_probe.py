#!/usr/bin/env python3
"""Quick single-GPU throughput probe (not the contract bench): python scripts/perf_probe.py [Mbp] [Mreads]"""
import os, sys, time, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from pymeme import hipapi, synth, workload

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 20
mreads = float(sys.argv[2]) if len(sys.argv) > 2 else 1
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 0
log = lambda s: print("[probe]", s, flush=True)
t0 = time.time()
g = synth.make_genome(int(mbp * 1e6), seed=11)
log("genome %.1f s" % (time.time() - t0))
tmp = tempfile.mkdtemp(prefix="probe_")
prefix = workload.build_index_on_disk(g, tmp, bits=bits, log=log)
ctx = hipapi.Context(0)
t0 = time.time(); ctx.load_index_files(prefix); log("index load+stage %.1f s" % (time.time() - t0))
n = int(mreads * 1e6)
t0 = time.time(); reads = workload.make_reads_fast(g, n, 150, seed=12); log("reads %.1f s" % (time.time() - t0))
d_reads = torch.from_numpy(reads.reshape(-1)).cuda()
d_off = torch.arange(0, (n + 1) * 150, 150, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
import itertools
configs = [(16,4),(8,4),(4,4),(8,8),(4,8),(16,8),(32,4)]
for lanes, bpc in configs:
    ctx.set_tuning("group_lanes", lanes); ctx.set_tuning("seed_blocks_per_cu", bpc)
    for it in range(2):
        res = ctx.seed_batch_device(d_reads.data_ptr(), d_off.data_ptr(), n, n * 150, hipapi.default_seed_opt(rounds=3))
        tm = ctx.timings()
    log("G=%d blocks/CU=%d: kernel %.1f ms pack %.2f ms gather %.1f ms -> %.2f M reads/s; windows/search %.2f" % (lanes, bpc, tm.seed_kernel_ms, tm.seed_pack_ms, tm.seed_gather_ms, n / tm.seed_kernel_ms / 1e3, tm.seed_windows / max(res.searches,1)))
ctx.set_tuning("group_lanes", 16); ctx.set_tuning("seed_blocks_per_cu", 4)
for rounds in (1, 2, 3):
    for it in range(2):
        t0 = time.time()
        res = ctx.seed_batch_device(d_reads.data_ptr(), d_off.data_ptr(), n, n * 150, hipapi.default_seed_opt(rounds=rounds))
        dt = time.time() - t0
        tm = ctx.timings()
    log("rounds=%d: %.3f s wall, kernel %.1f ms (+gather %.1f ms, launches %d) -> %.2f M reads/s (kernel); smems/read %.2f hits/read %.2f searches/read %.1f" % (
        rounds, dt, tm.seed_kernel_ms, tm.seed_gather_ms, tm.seed_launches, n / tm.seed_kernel_ms / 1e3, res.total_smems / n, res.total_hits / n, res.searches / n))
# BSW probe
import bsw_gen
pairs, ref, qer = bsw_gen.make_pairs(20000, seed=3)
reps = 50
P = np.tile(pairs, reps); 
d_pairs = torch.from_numpy(P.view(np.uint8)).cuda(); d_ref = torch.from_numpy(ref).cuda(); d_qer = torch.from_numpy(qer).cuda()
torch.cuda.synchronize()
for it in range(2):
    ctx.bsw_batch_device(d_pairs.data_ptr(), d_ref.data_ptr(), d_qer.data_ptr(), P.shape[0], 100)
tm = ctx.timings()
log("bsw: %d pairs in %.1f ms -> %.2f M pairs/s" % (P.shape[0], tm.bsw_kernel_ms, P.shape[0] / tm.bsw_kernel_ms / 1e3))
