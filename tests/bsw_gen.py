"""Synthetic seed-extension tasks shaped like the ones mem_chain2aln_across_reads_V2 builds
(reference src/bwamem.cpp:2733-2914): a query tail of a read and a somewhat longer reference window,
related by substitutions and small indels, with the running score h0 of the seed."""
import numpy as np

from oracle_py import SEQPAIR_DTYPE


def make_pairs(n, seed=0, max_q=150, sub=0.03, indel=0.01, amb=0.002, h0_max=120, eight_bit=False,
               min_q=1, unrelated_frac=0.05):
    rng = np.random.default_rng(seed)
    pairs = np.zeros(n, dtype=SEQPAIR_DTYPE)
    refs, qers = [], []
    ro = qo = 0
    for i in range(n):
        ql = int(rng.integers(min_q, max_q + 1))
        if eight_bit:
            ql = min(ql, 120)
        q = rng.integers(0, 4, size=ql).astype(np.uint8)
        # derive the reference from the query with edits, then append flanking sequence
        if rng.random() < unrelated_frac:
            t = rng.integers(0, 4, size=ql + int(rng.integers(0, 40))).astype(np.uint8)
        else:
            t = []
            for b in q:
                r = rng.random()
                if r < indel / 2:
                    continue  # deletion from the reference
                if r < indel:
                    t.extend(rng.integers(0, 4, size=int(rng.integers(1, 6))))
                t.append(int(b) if rng.random() >= sub else int((b + rng.integers(1, 4)) & 3))
            t.extend(rng.integers(0, 4, size=int(rng.integers(0, 60))))
            t = np.asarray(t, dtype=np.uint8)
        if eight_bit:
            t = t[:120]
        if t.shape[0] == 0:
            t = rng.integers(0, 4, size=1).astype(np.uint8)
        am = rng.random(t.shape[0]) < amb
        t[am] = 4
        am = rng.random(ql) < amb
        q[am] = 4
        h0 = int(rng.integers(1, h0_max + 1))
        if eight_bit:
            h0 = max(1, min(h0, 127 - min(ql, t.shape[0]) - 1))
        pairs[i]["idr"], pairs[i]["idq"] = ro, qo
        pairs[i]["len1"], pairs[i]["len2"] = t.shape[0], ql
        pairs[i]["h0"] = h0
        pairs[i]["seqid"], pairs[i]["regid"], pairs[i]["id"] = i, 0, i
        refs.append(t)
        qers.append(q)
        ro += t.shape[0]
        qo += ql
    return pairs, np.concatenate(refs), np.concatenate(qers)


OUT_FIELDS = ("score", "tle", "gtle", "qle", "gscore", "max_off")


def outputs(pairs):
    return np.stack([pairs[f] for f in OUT_FIELDS], axis=1)
