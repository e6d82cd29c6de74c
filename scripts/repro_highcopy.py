import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import oracle_py as O
from pymeme import hipapi, hostapi, synth, workload
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 8
g = synth.make_genome(int(mbp * 1e6), seed=11, repeat_frac=0.2, n_families=2, divergence=0.03)
text, sa = hostapi.build_sa(g); l1, l2 = hostapi.train_prmi(text, sa, bits=16)
print("partial models", l1.shape[0], flush=True)
ctx = hipapi.Context(0)
pp = np.zeros((sa.shape[0], 5), np.uint8); pp[:, :4] = (sa >> np.uint64(8)).astype('<u4').view(np.uint8).reshape(-1, 4); pp[:, 4] = (sa & np.uint64(255)).astype(np.uint8)
ctx.load_index_host(pp.reshape(-1), text, l1, l2)
n = 20000
reads = workload.make_reads_fast(g, n, 150, seed=12)
off = np.arange(0, (n + 1) * 150, 150, dtype=np.int64)
t = time.time()
smems, smem_off, hits, hit_off = ctx.seed_batch(reads, off, hipapi.default_seed_opt(rounds=3, hits_per_smem=0))
print("gpu ok", time.time() - t, smems.shape, hits.shape, np.diff(smem_off).max(), flush=True)
idx = O.Index(text, sa)
bad = 0
for lo in range(0, n, 2000):
    sub = reads[lo:lo + 2000]; o2 = np.arange(0, 2001 * 150, 150, dtype=np.int64)
    sm, ns, oh, nh, _ = O.seed_batch(idx, sub, o2, smem_cap=4096, hit_cap=1 << 17, threads=0)
    for r in range(2000):
        a = smems[smem_off[lo + r]:smem_off[lo + r + 1]]
        k = ns[r]
        if a.shape[0] != k or not all(np.array_equal(np.sort(a[f]), np.sort(sm[r, :k][f])) for f in ("start", "end", "hitcount")):
            bad += 1
            if bad < 4: print("mismatch read", lo + r, a.shape[0], k)
print("mismatching reads:", bad)
