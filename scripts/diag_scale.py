#!/usr/bin/env python3
"""Large-index diagnostic: python scripts/diag_scale.py <Mbp> -- builds the index on the host, checks the suffix
array order on a sample, seeds read chunks round by round and isolates reads that misbehave."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch, ctypes as C
import oracle_py as O
from pymeme import hipapi, hostapi, synth, workload
log = lambda *a: print("[diag]", *a, flush=True)
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 2200
l_pac = int(mbp * 1e6) & ~1; n = 2 * l_pac
t0 = time.time(); g = synth.make_genome(l_pac, seed=11); text, sa = hostapi.build_sa(g); log("sa built", time.time() - t0)
t0 = time.time(); l1, l2 = hostapi.train_prmi(text, sa); log("rmi", time.time() - t0, l1.shape, l2.shape)
# SA order check on a sample
rng = np.random.default_rng(0)
idx = O.Index(text, sa)
tp = np.concatenate([text, np.full(300, 3, np.uint8)])
bad = 0
for i in rng.integers(0, n - 1, 200000):
    a, b = int(sa[i]), int(sa[i + 1])
    x, y = tp[a:a + 200], tp[b:b + 200]
    k = min(x.shape[0], y.shape[0]); d = np.nonzero(x[:k] != y[:k])[0]
    if d.size and x[d[0]] > y[d[0]]: bad += 1
log("SA order violations in sample:", bad, "max sa", int(sa.max()), "n", n)
dev = torch.device("cuda", 0)
d_text = torch.from_numpy(text).to(dev); d_sa = torch.from_numpy(sa.view(np.int64)).to(dev)
d_l2 = torch.from_numpy(l2.view(np.uint8).reshape(-1)).to(dev); d_l1 = torch.from_numpy(np.ascontiguousarray(l1).view(np.uint8).reshape(-1)).to(dev) if l1.shape[0] else torch.zeros(24, dtype=torch.uint8, device=dev)
ctx = hipapi.Context(0); L = hipapi.lib()
words = L.meme_index_pac64_words(n)
d_pac = torch.empty(words, dtype=torch.int64, device=dev); d_ent = torch.empty(2 * n, dtype=torch.int64, device=dev)
hipapi._check(L.meme_stage_pack_text(C.c_void_p(ctx.h), C.c_void_p(d_text.data_ptr()), C.c_int64(n), C.c_void_p(d_pac.data_ptr())))
hipapi._check(L.meme_stage_entries_from_sa(C.c_void_p(ctx.h), C.c_void_p(d_sa.data_ptr()), C.c_int64(n), C.c_void_p(d_pac.data_ptr()), C.c_void_p(d_ent.data_ptr())))
ctx.sync()
ctx.attach_index(hipapi.IndexArrays(n, d_ent.data_ptr(), d_pac.data_ptr(), d_l2.data_ptr(), l2.shape[0], d_l1.data_ptr(), l1.shape[0]))
# entries check: keys non-decreasing on a sample, pos equals sa
ent = d_ent.view(-1, 2)
for lo in (0, n // 2, n - 1000000):
    e = ent[lo:lo + 1000000].cpu().numpy()
    keys = e[:, 0].view(np.uint64); pos = e[:, 1].view(np.uint64)
    log("entries @", lo, "keys sorted:", bool(np.all(keys[1:] >= keys[:-1])), "pos ok:", bool(np.array_equal(pos, sa[lo:lo + 1000000])))
nreads = 200000
reads = workload.make_reads_fast(g, nreads, 150, seed=12)
off = np.arange(0, (nreads + 1) * 150, 150, dtype=np.int64)
for rounds in (1, 2, 3):
    for lo in range(0, nreads, 50000):
        sub = reads[lo:lo + 50000]; o2 = np.arange(0, 50001 * 150, 150, dtype=np.int64)
        try:
            t0 = time.time()
            smems, so, hits, ho = ctx.seed_batch(sub, o2, hipapi.default_seed_opt(rounds=rounds, hits_per_smem=4))
            tm = ctx.timings()
            log("rounds", rounds, "chunk", lo, "ok: max smems/read", int(np.diff(so).max()), "launches", tm.seed_launches, "kernel ms %.1f" % tm.seed_kernel_ms, "smems", smems.shape[0])
        except Exception as e:
            log("rounds", rounds, "chunk", lo, "FAILED", e)
            # isolate
            for r in range(50000):
                try:
                    ctx.seed_batch(sub[r:r + 1], np.array([0, 150], np.int64), hipapi.default_seed_opt(rounds=rounds, hits_per_smem=4))
                except Exception as e2:
                    log("  read", lo + r, "fails:", e2, "".join("ACGTN"[c] for c in sub[r]))
                    sm, ns, oh, nh, ctr = O.seed_batch(idx, sub[r:r + 1], np.array([0, 150], np.int64), O.default_seed_params(steps=rounds), smem_cap=1 << 17, hit_cap=1 << 22, threads=1)
                    log("  oracle: smems", int(ns[0]), "hits", int(nh[0]), "searches", ctr.searches)
                    break
            break
# parity vs oracle on 3000 reads
sub = reads[:3000]; o2 = np.arange(0, 3001 * 150, 150, dtype=np.int64)
try:
    smems, so, hits, ho = ctx.seed_batch(sub, o2, hipapi.default_seed_opt(rounds=3, hits_per_smem=0))
    sm, ns, oh, nh, _ = O.seed_batch(idx, sub, o2, smem_cap=4096, hit_cap=1 << 17, threads=0)
    slots, counts, hl = hipapi.smems_to_slots(smems, so, hits, ho)
    log("parity vs oracle on 3000 reads:", O.format_seed_dump(slots, counts, hl) == O.format_seed_dump(sm, ns, oh))
except Exception as e:
    log("parity run failed", e)
