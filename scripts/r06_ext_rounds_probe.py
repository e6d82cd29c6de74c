#!/usr/bin/env python3
"""Round 6 probe (reference src/bwamem.cpp:3389-3485, 1680-1693): on a repeat-dense genome, the extension stage in rounds (surviving records only)
against the stage in the reference's batch order minus the purged records -- where and how do they differ?   python scripts/r06_ext_rounds_probe.py [Mbp] [reads]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from pymeme import hipapi, workload
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 64
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
l_pac = int(mbp * 1e6) & ~1; n = 2 * l_pac
g = workload.repeat_dense_genome(l_pac)
text = hipapi.fwd_rc_text(g)
ctx = hipapi.Context(0)
d_text, d_sa = hipapi.build_sa_device(ctx, text)
d_pos5 = hipapi.pos5_from_sa_torch(ctx, d_sa, n); del d_sa
d_pac, d_ent = hipapi.stage_entries_torch(ctx, n, d_text, d_pos5)
d_l2, n_l2, d_l1, n_l1 = hipapi.train_prmi_device(ctx, d_ent, n, 24)
keep = (d_pac, d_ent) + hipapi.attach_index_torch(ctx, n, d_pac, d_ent, d_l2, n_l2, d_l1, n_l1)
L = 150
reads = workload.make_reads_fast(g, nreads, L, seed=78)
off = np.arange(0, (nreads + 1) * L, L, dtype=np.int64)
ctx.seed_batch_host(reads.reshape(-1), off)
contigs = [(l_pac * i // 4, l_pac * (i + 1) // 4 - l_pac * i // 4, 0) for i in range(4)]
copt = hipapi.default_chain_opt(l_pac)
R = ctx.extend_last_batch_host(contigs, copt)
regs, ro = R["regs"].copy(), R["reg_off"].copy()
for rounds in [int(x) for x in os.environ.get("PROBE_ROUNDS", "1,0,2").split(",")]:
    ctx.set_tuning("ext_live_only", 1); ctx.set_tuning("ext_rounds", rounds)
    RL = ctx.extend_last_batch_host(contigs, copt)
    ctx.set_tuning("ext_live_only", 0)
    lregs, lro = RL["regs"].copy(), RL["reg_off"].copy()
    keepm = regs["qe"] > regs["qb"]
    want_off = np.concatenate([[0], np.cumsum(keepm.astype(np.int64))])[ro]
    same_off = np.array_equal(lro, want_off)
    want = regs[keepm]
    print("ext_rounds=%d: records %d (expected %d), offsets equal %s, bytes equal %s" % (rounds, lregs.shape[0], want.shape[0], same_off, hipapi.records_equal(lregs, want)), flush=True)
    bad = np.nonzero(np.diff(lro) != np.diff(want_off))[0]
    print("  reads whose number of surviving records differs: %d" % bad.shape[0])
    if same_off:
        for f in lregs.dtype.names:
            d = np.nonzero(lregs[f] != want[f])[0]
            if d.shape[0]:
                print("  field %s differs in %d records; first: rounds %r vs batch order %r" % (f, d.shape[0], lregs[f][d[0]], want[f][d[0]]))
    for r in bad[:3]:
        print("  read %d: batch order (all records: qb qe rb re score w seedcov):" % r)
        for x in regs[ro[r]:ro[r + 1]]:
            print("     ", int(x["qb"]), int(x["qe"]), int(x["rb"]), int(x["re"]), int(x["score"]), int(x["w"]), int(x["seedcov"]), int(x["seedlen0"]))
        print("   in rounds:")
        for x in lregs[lro[r]:lro[r + 1]]:
            print("     ", int(x["qb"]), int(x["qe"]), int(x["rb"]), int(x["re"]), int(x["score"]), int(x["w"]), int(x["seedcov"]), int(x["seedlen0"]))
ctx.close()
