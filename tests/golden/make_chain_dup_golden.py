#!/usr/bin/env python3
"""Generates tests/golden/chain_dup_golden.npz: made-up SMEM + hit sets (tests/chain_gen.py: chains at EQUAL positions, hundreds of chains
per read, repeat SMEMs beyond max_occ) and the chains the COMPILED REFERENCE's own mem_chain_Learned + mem_chain_flt (klib B-tree and
introsort included) make of them, through oracle/_ref/libstage_ref.so (oracle/ref_stage_shim.cpp).  Runs in the build container (no GPU).
Data only: inputs and the reference's outputs."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import chain_gen  # noqa: E402
import oracle_py as O  # noqa: E402
import ref_py  # noqa: E402

L_PAC = 200_000
CONTIG_OFF = np.array([0, 70_000, 150_000], np.int64)
CONTIG_LEN = np.array([70_000, 80_000, 50_000], np.int32)
CONTIG_ALT = np.array([0, 0, 1], np.uint8)


def main(out, n=300, seed=4242):
    opt = O.default_chain_opt(L_PAC)
    reads = chain_gen.workload(seed, n, l_pac=L_PAC)
    smem_off = np.zeros(n + 1, np.int64); hit_off = np.zeros(n + 1, np.int64); chain_off = np.zeros(n + 1, np.int64); seed_off = np.zeros(n + 1, np.int64)
    smems, hits, chains, seeds, tree, frac, rlen = [], [], [], [], [], [], []
    for r, (sm, h) in enumerate(reads):
        L = 250 if r % 5 == 2 else 150
        rc, ch, sd, t, f = ref_py.chain_read(sm, h, L, CONTIG_OFF, CONTIG_LEN, CONTIG_ALT, opt)
        assert rc >= 0
        smems.append(np.stack([sm["start"], sm["end"], sm["hitbeg"], sm["hitcount"]], 1).astype(np.int32)); hits.append(h)
        chains.append(np.stack([ch[k] for k in ("pos", "rid", "n_seeds", "w", "kept", "first", "is_alt", "seed_beg")], 1).astype(np.int64))
        seeds.append(np.stack([sd["rbeg"], sd["qbeg"], sd["len"]], 1).astype(np.int64))
        tree.append(t); frac.append(np.float32(f).view(np.uint32)); rlen.append(L)
        smem_off[r + 1] = smem_off[r] + sm.shape[0]; hit_off[r + 1] = hit_off[r] + h.shape[0]
        chain_off[r + 1] = chain_off[r] + rc; seed_off[r + 1] = seed_off[r] + sd.shape[0]
    np.savez_compressed(out, read_len=np.array(rlen, np.int32), tree_size=np.array(tree, np.int32), frac_rep_bits=np.array(frac, np.uint32),
                        smem_off=smem_off, smems=np.concatenate(smems), hit_off=hit_off, hits=np.concatenate(hits), chain_off=chain_off,
                        chains=np.concatenate(chains), seed_off=seed_off, seeds=np.concatenate(seeds), l_pac=np.int64(L_PAC),
                        contig_off=CONTIG_OFF, contig_len=CONTIG_LEN, contig_alt=CONTIG_ALT)
    print("reads", n, "chains", int(chain_off[-1]), "seeds", int(seed_off[-1]), "max tree", max(tree))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "chain_dup_golden.npz"))
