#!/usr/bin/env python3
"""Generates tests/golden/chain_golden.npz: per read the seeds (SMEMs + hits, as the HIP backend delivers them -- equal to the
reference's, tests/test_gpu_seed.py) and the chains the COMPILED REFERENCE's own mem_chain_Learned + mem_chain_flt make of them.

Needs a GPU box with oracle/_ref (the reference aligner bound to the backend dumps what the reference's host functions
compute, MEME_DROPIN_CHAIN_DUMP):   python tests/golden/make_chain_golden.py <output.npz>
Data only: inputs and the reference's outputs."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))

from common import build_index  # noqa: E402
from pymeme import synth  # noqa: E402


def main(out):
    d = tempfile.mkdtemp(prefix="chain_golden_")
    g = synth.make_genome(300_000, seed=201, repeat_frac=0.15, repeat_len=250, n_families=5, divergence=0.02, n_dups=8, dup_len=1200, poly_runs=4)
    fa = os.path.join(d, "c.fa")
    synth.write_fasta(fa, g, name="cg", contigs=3)
    prefix = build_index(fa, bits=14)
    r1, _, _ = synth.make_reads(g, 2000, 150, seed=202, n_frac=0.03, exact_frac=0.2)
    r2, _, _ = synth.make_reads(g, 500, 250, seed=203, sub_rate=0.05, indel_rate=0.0075, n_frac=0.02)
    r3, _, _ = synth.make_reads(g, 300, 60, seed=204, sub_rate=0.02)
    fq = os.path.join(d, "c.fq")
    with open(fq, "w") as fh:
        k = 0
        for rs in (r1, r2, r3):
            for r in rs:
                L = len(r) if rs is not r3 else 15 + k % 46
                fh.write("@c%d\n%s\n+\n%s\n" % (k, "".join("ACGTN"[c] for c in r[:L]), "I" * L))
                k += 1
    dump = os.path.join(d, "dump.txt")
    env = dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_CHAIN_DUMP=dump, MEME_DROPIN_CHAIN_CHECK="1")
    r = subprocess.run([os.path.join(REPO, "oracle", "_ref", "bwa-meme_dropin"), "mem", "-7", "-Y", "-t", "8", prefix, fq], stdout=subprocess.DEVNULL,
                       stderr=subprocess.PIPE, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    recs = {}
    cur = None
    for line in open(dump):
        t = line.split()
        if t[0] == "R":
            cur = {"len": int(t[2]), "tree": int(t[5]), "frac": int(t[7]), "S": [], "H": [], "C": []}
            recs[int(t[1])] = cur
        elif t[0] == "S":
            cur["S"].append([int(x) for x in t[1:5]])
        elif t[0] == "H":
            cur["H"] = [int(x) for x in t[1:]]
        elif t[0] == "C":
            i = t.index(":")
            cur["C"].append(([int(x) for x in t[1:i]], [int(x) for x in t[i + 1:]]))
    n = len(recs)
    assert sorted(recs) == list(range(n)), "reads missing from the dump"
    read_len = np.array([recs[i]["len"] for i in range(n)], np.int32)
    tree = np.array([recs[i]["tree"] for i in range(n)], np.int32)
    frac = np.array([recs[i]["frac"] for i in range(n)], np.uint32)
    smem_off = np.zeros(n + 1, np.int64); hit_off = np.zeros(n + 1, np.int64); chain_off = np.zeros(n + 1, np.int64)
    smems, hits, chains, seeds = [], [], [], []
    for i in range(n):
        rc = recs[i]
        smems += rc["S"]; hits += rc["H"]
        for hdr, sd in rc["C"]:
            chains.append(hdr + [len(seeds)])
            seeds += [sd[3 * j:3 * j + 3] for j in range(len(sd) // 3)]
        smem_off[i + 1] = len(smems); hit_off[i + 1] = len(hits); chain_off[i + 1] = len(chains)
    ann = [l.split() for l in open(prefix + ".ann")]
    l_pac = int(ann[0][0])
    contig_off, contig_len = [], []
    for k in range(int(ann[0][1])):
        contig_off.append(int(ann[2 + 2 * k][0])); contig_len.append(int(ann[2 + 2 * k][1]))
    np.savez_compressed(out, read_len=read_len, tree_size=tree, frac_rep_bits=frac, smem_off=smem_off, smems=np.array(smems, np.int32).reshape(-1, 4),
                        hit_off=hit_off, hits=np.array(hits, np.uint64), chain_off=chain_off,
                        chains=np.array(chains, np.int64).reshape(-1, 8),      # pos rid n w kept first is_alt seed_beg(global)
                        seeds=np.array(seeds, np.int64).reshape(-1, 3), l_pac=np.int64(l_pac), contig_off=np.array(contig_off, np.int64),
                        contig_len=np.array(contig_len, np.int32),
                        opt=np.array([100, 10000, 500, 19, 0, 1 << 30], np.int32), opt_f=np.array([0.5, 0.5], np.float32))
    print("reads", n, "chains", len(chains), "seeds", len(seeds), "reads with >1 chain", int((np.diff(chain_off) > 1).sum()))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "chain_golden.npz"))
