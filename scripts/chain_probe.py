#!/usr/bin/env python3
"""Chaining-on-the-device probe: python scripts/chain_probe.py [Mbp] [Mreads]
Seeds a batch through the pinned-result call, chains it with meme_chain_last_batch_host and prints wall times of both calls
(kernel times: run under `rocprofv3 --kernel-trace --stats`)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))
import numpy as np, torch
from pymeme import hipapi, synth, workload
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 256
nreads = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 2000000
l_pac = int(mbp * 1e6) & ~1
n = 2 * l_pac
g = synth.make_genome(l_pac, seed=11)
ctx = hipapi.Context(0)
text = hipapi.fwd_rc_text(g)
d_text, d_sa = hipapi.build_sa_device(ctx, text)
d_pos5 = hipapi.pos5_from_sa_torch(ctx, d_sa, n)
del d_sa
d_pac, d_ent = hipapi.stage_entries_torch(ctx, n, d_text, d_pos5)
bits = 28 if 8.0 * n + 8 > 8.0e9 else 26 if 8.0 * n + 8 > 1.0e9 else 24
d_l2, n_l2, d_l1, n_l1 = hipapi.train_prmi_device(ctx, d_ent, n, bits)
keep = (d_pac, d_ent) + hipapi.attach_index_torch(ctx, n, d_pac, d_ent, d_l2, n_l2, d_l1, n_l1)
reads = workload.make_reads_fast(g, nreads, 150, seed=1000)
off = np.arange(0, (nreads + 1) * 150, 150, dtype=np.int64)
contigs = [(l_pac * k // 8, l_pac * (k + 1) // 8 - l_pac * k // 8, 0) for k in range(8)]
opt = hipapi.default_chain_opt(l_pac)
if os.environ.get("CHAIN_LIGHT_HITS"): ctx.set_tuning("chain_light_hits", int(os.environ["CHAIN_LIGHT_HITS"]))
if os.environ.get("CHAIN_WAVE_TIERS"): ctx.set_tuning("chain_wave_tiers", int(os.environ["CHAIN_WAVE_TIERS"]))
if os.environ.get("CHAIN_SIDE_PRIORITY"): ctx.set_tuning("chain_side_priority", int(os.environ["CHAIN_SIDE_PRIORITY"]))
for it in range(3 if not os.environ.get("CHAIN_LANE_HITS") else 1):
    t0 = time.time(); smems, so, hits, ho = ctx.seed_batch_host(reads.reshape(-1), off); t1 = time.time()
    res = ctx.chain_last_batch_host(contigs, opt); t2 = time.time()
    print("[chain probe] %d reads: seeding call %.1f ms (%.1f M reads/s incl. transfers), chaining call %.1f ms (%.1f M reads/s incl. transfers + numpy copies); "
          "%d chains, %d chained seeds, %d reads left to the host" % (nreads, (t1 - t0) * 1e3, nreads / (t1 - t0) / 1e6, (t2 - t1) * 1e3, nreads / (t2 - t1) / 1e6,
                                                                      res["chains"].shape[0], res["seeds"].shape[0], res["n_fallback"]), flush=True)
# ---- distribution of the per-read work (what the second pass has to cope with) ----
hc = np.minimum(smems["hitcount"].astype(np.int64), 500)
cs = np.concatenate([[0], np.cumsum(hc)])
work = cs[so[1:]] - cs[so[:-1]]
ns = np.diff(so)
tree = res["tree_size"]
def pct(a):
    return " ".join("%s=%d" % (p, np.percentile(a, p)) for p in (50, 90, 99, 99.9, 99.99, 100))
print("[chain probe] work per read (sampled hits walked): " + pct(work))
print("[chain probe] SMEMs per read: " + pct(ns))
print("[chain probe] chains before the filter: " + pct(tree))
for lim in (16, 32, 64, 128, 256, 1024):
    print("[chain probe] reads with more than %d chains: %d; with work > %d: %d" % (lim, int((tree > lim).sum()), lim * 8, int((work > lim * 8).sum())))
try:                                    # investigation builds (-DMEME_CHAIN_PROF): cycles per phase of the wavefront tiers, all launches so far
    import ctypes
    f = hipapi.lib().meme_debug_chain_prof
    buf = (ctypes.c_ulonglong * 32)()
    if f(buf, 0) == 0:
        names = ["", "", "", "", "", "", "", "",
                 "lds: smem order", "lds: walk", "lds: records + rank", "lds: -", "lds: sort", "lds: filter", "lds: marks + output"]
        for k, nm in enumerate(names):
            if nm: print("[chain probe] phase %-22s %8.1f Mcycles" % (nm, buf[k] / 1e6))
except AttributeError:
    pass
tm = ctx.timings()
print("[chain probe] chain kernels %.2f ms, of which the wavefront tiers %.2f ms (%d reads), of which the B-tree tier %.2f ms (%d reads)"
      % (tm.chain_kernel_ms, tm.chain_pass2_ms, tm.chain_tier2_reads, tm.chain_tier3_ms, tm.chain_tier3_reads))
big = np.argsort(work)[-10:]
print("[chain probe] ten heaviest reads: work", work[big].tolist(), "smems", ns[big].tolist(), "chains", tree[big].tolist())

for cap in [int(x) for x in os.environ.get("CHAIN_LANE_HITS", "").split(",") if x]:
    ctx.set_tuning("chain_lane_hits", cap)
    for _ in range(2):
        res = ctx.chain_last_batch_host(contigs, opt)
    tm = ctx.timings()
    print("[chain probe] lane tier walks <= %d hits: chain kernels %.2f ms, wavefront tiers %.2f ms (%d reads), B-tree tier %.2f ms (%d reads)"
          % (cap, tm.chain_kernel_ms, tm.chain_pass2_ms, tm.chain_tier2_reads, tm.chain_tier3_ms, tm.chain_tier3_reads))

for lh in [int(x) for x in os.environ.get("CHAIN_LIGHT_SWEEP", "").split(",") if x]:
    ctx.set_tuning("chain_light_hits", lh)
    for _ in range(3):
        res = ctx.chain_last_batch_host(contigs, opt)
    tm = ctx.timings()
    print("[chain probe] reads with more than %d hits skip the lane tier: chain kernels %.2f ms, after the lane tier %.2f ms (%d reads in the wavefront tiers)"
          % (lh, tm.chain_kernel_ms, tm.chain_pass2_ms, tm.chain_tier2_reads))
