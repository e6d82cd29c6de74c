"""Synthetic workloads for bench.py / scale tests: genome + index (host builder) and device-resident reads."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile
import time

import numpy as np

from . import synth

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_index_on_disk(genome: np.ndarray, workdir: str, bits: int = 0, threads: int = 0, contigs: int = 4,
                        log=None) -> str:
    """FASTA -> our meme-index (reference file formats).  Returns the index prefix."""
    fa = os.path.join(workdir, "ref.fa")
    t0 = time.time()
    synth.write_fasta(fa, genome, contigs=contigs)
    cmd = [os.path.join(PKG, "meme-index"), "build", fa]
    if bits:
        cmd += ["-b", str(bits)]
    if threads:
        cmd += ["-t", str(threads)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("meme-index failed: " + r.stderr[-2000:])
    if log:
        log("index build %.1f s: %s" % (time.time() - t0, r.stderr.strip().replace("\n", " | ")))
    return fa


def make_reads_fast(genome: np.ndarray, n_reads: int, read_len: int, seed: int, sub_rate=0.01, n_frac=0.02,
                    indel_frac=0.15, chunk=1 << 20) -> np.ndarray:
    """Vectorised read sampler for millions of reads: both strands, substitutions, one short indel in
    `indel_frac` of the reads, one N in `n_frac` of the reads.  Returns codes [n_reads, read_len]."""
    rng = np.random.default_rng(seed)
    n = genome.shape[0]
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    span = read_len + 4
    ar = np.arange(span)
    for s in range(0, n_reads, chunk):
        m = min(chunk, n_reads - s)
        pos = rng.integers(0, n - span, size=m)
        frag = genome[pos[:, None] + ar[None, :]]
        sub = rng.random((m, span)) < sub_rate
        frag = np.where(sub, (frag + rng.integers(1, 4, size=(m, span), dtype=np.uint8)) & 3, frag).astype(np.uint8)
        # deletion of 1..3 bases at a random column for a subset: shift the tail left
        has = rng.random(m) < indel_frac
        col = rng.integers(10, read_len - 10, size=m)
        dl = rng.integers(1, 4, size=m)
        idx = ar[None, :read_len] + np.where((ar[None, :read_len] >= col[:, None]) & has[:, None], dl[:, None], 0)
        reads = np.take_along_axis(frag, idx, axis=1)
        rcm = rng.random(m) < 0.5
        reads[rcm] = 3 - reads[rcm][:, ::-1]
        wn = np.nonzero(rng.random(m) < n_frac)[0]
        reads[wn, rng.integers(0, read_len, size=wn.size)] = 4
        out[s:s + m] = reads
    return out


def make_bsw_pairs(n_pairs: int, seed: int, read_len: int = 150, base: int = 8192, sub=0.01, indel=0.0015,
                   amb=0.001):
    """Seed-extension jobs shaped like the ones mem_chain2aln_across_reads_V2 builds for `read_len`-bp reads
    (reference src/bwamem.cpp:2777-2798, 2893-2898): the query is the part of the read left or right of a seed
    (1 .. read_len-19 bases), the target the same stretch of the reference plus the gap allowance, h0 the
    seed's score.  `base` distinct pairs are generated and tiled to n_pairs (the work per pair is what matters).
    Returns (pairs[SEQPAIR], ref bytes, query bytes)."""
    from .hipapi import SEQPAIR
    rng = np.random.default_rng(seed)
    base = min(base, n_pairs)
    pairs = np.zeros(base, dtype=SEQPAIR)
    refs, qers = [], []
    ro = qo = 0
    for i in range(base):
        ql = int(rng.integers(1, read_len - 19 + 1))
        q = rng.integers(0, 4, size=ql).astype(np.uint8)
        t = q.copy()
        sm = rng.random(ql) < sub
        t[sm] = (t[sm] + rng.integers(1, 4, size=int(sm.sum()))) & 3
        if rng.random() < indel * ql:                       # one short indel
            p = int(rng.integers(0, ql))
            k = int(rng.integers(1, 4))
            t = np.concatenate([t[:p], t[p + k:]]) if rng.random() < 0.5 else \
                np.concatenate([t[:p], rng.integers(0, 4, size=k).astype(np.uint8), t[p:]])
        gap = int(rng.integers(10, 60))                     # cal_max_gap allowance (src/bwamem.cpp:85-95)
        t = np.concatenate([t, rng.integers(0, 4, size=gap).astype(np.uint8)])
        t[rng.random(t.shape[0]) < amb] = 4
        pairs[i]["idr"], pairs[i]["idq"] = ro, qo
        pairs[i]["len1"], pairs[i]["len2"] = t.shape[0], ql
        pairs[i]["h0"] = int(read_len - ql) if rng.random() < 0.7 else int(rng.integers(19, read_len))
        refs.append(t); qers.append(q)
        ro += t.shape[0]; qo += ql
    reps = (n_pairs + base - 1) // base
    out = np.tile(pairs, reps)[:n_pairs].copy()
    out["id"] = np.arange(n_pairs, dtype=np.int32)
    out["seqid"] = out["id"]
    return out, np.concatenate(refs), np.concatenate(qers), base


def _bsw_pairs_distinct_block(n_pairs: int, rng, read_len, sub, indel_frac, amb):
    ql = rng.integers(1, read_len - 19 + 1, size=n_pairs).astype(np.int32)
    gap = rng.integers(10, 60, size=n_pairs).astype(np.int32)
    ind = np.where(rng.random(n_pairs) < indel_frac, rng.integers(-3, 4, size=n_pairs), 0).astype(np.int32)
    ind = np.where(ql + ind < 1, 0, ind).astype(np.int32)
    tl = ql + ind + gap
    qo = np.zeros(n_pairs + 1, np.int64); qo[1:] = np.cumsum(ql)
    to = np.zeros(n_pairs + 1, np.int64); to[1:] = np.cumsum(tl)
    qer = rng.integers(0, 4, size=int(qo[-1]), dtype=np.uint8)
    ref = rng.integers(0, 4, size=int(to[-1]), dtype=np.uint8)
    # copy the query into the head of its target, shifted past the indel point
    pid_q = np.repeat(np.arange(n_pairs, dtype=np.int32), ql)
    j = (np.arange(int(qo[-1]), dtype=np.int64) - qo[pid_q]).astype(np.int32)     # position inside the query
    tj = j + np.where(j >= (ql >> 1)[pid_q], ind[pid_q], 0).astype(np.int32)
    ok = (tj >= 0) & (tj < (ql + ind)[pid_q])
    ref[(to[pid_q] + tj)[ok]] = qer[ok]
    # substitutions and ambiguous bases at their expected number of random positions (cheaper than a draw per base)
    nb = ref.shape[0]
    sm = rng.integers(0, nb, size=rng.binomial(nb, sub))
    ref[sm] = (ref[sm] + rng.integers(1, 4, size=sm.shape[0], dtype=np.uint8)) & 3
    ref[rng.integers(0, nb, size=rng.binomial(nb, amb))] = 4
    h0 = np.where(rng.random(n_pairs) < 0.7, read_len - ql, rng.integers(19, read_len, size=n_pairs))
    return ql, tl, qo, to, qer, ref, h0


def make_bsw_pairs_distinct(n_pairs: int, seed: int, read_len: int = 150, sub=0.01, indel_frac=0.2, amb=0.001):
    """Like make_bsw_pairs, but every pair is its own job (vectorised generator, built in cache-sized blocks): query =
    1 .. read_len-19 random bases, target = the same stretch with substitutions, one 1-3 base indel in `indel_frac` of the
    pairs, plus the gap allowance (10-59 random bases, cal_max_gap, reference src/bwamem.cpp:85-95); h0 the seed's score.
    Sequences are packed back to back.  Returns (pairs[SEQPAIR], ref bytes, query bytes)."""
    from .hipapi import SEQPAIR
    rng = np.random.default_rng(seed)
    pairs = np.zeros(n_pairs, dtype=SEQPAIR)
    refs, qers = [], []
    rbase = qbase = 0
    block = 1 << 16
    for p0 in range(0, n_pairs, block):
        m = min(block, n_pairs - p0)
        ql, tl, qo, to, qer, ref, h0 = _bsw_pairs_distinct_block(m, rng, read_len, sub, indel_frac, amb)
        v = pairs[p0:p0 + m]
        v["idr"] = to[:-1] + rbase; v["idq"] = qo[:-1] + qbase
        v["len1"] = tl; v["len2"] = ql; v["h0"] = h0
        refs.append(ref); qers.append(qer)
        rbase += ref.shape[0]; qbase += qer.shape[0]
    pairs["id"] = np.arange(n_pairs, dtype=np.int32)
    pairs["seqid"] = pairs["id"]
    return pairs, np.concatenate(refs), np.concatenate(qers)


def write_fastq_fast(path: str, reads: np.ndarray, prefix: str = "r", first: int = 0, append: bool = False) -> None:
    """Vectorised FASTQ writer for millions of fixed-length reads (names <prefix><first + index>, constant qualities): reads whose
    numbers have the same count of digits form one fixed-width byte matrix."""
    n, L = reads.shape
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    pre = np.frombuffer(b"@" + prefix.encode(), dtype=np.uint8)
    with open(path, "ab" if append else "wb") as fh:
        step = 1 << 20
        for s0 in range(0, n, step):
            ids = np.arange(first + s0, first + min(n, s0 + step), dtype=np.int64)
            nd = np.where(ids > 0, np.floor(np.log10(np.maximum(ids, 1))).astype(np.int64) + 1, 1)
            nd = np.where(10 ** (nd - 1) > ids, nd - 1, nd)                       # (guards against log10 rounding at powers of ten)
            nd = np.maximum(np.where(10 ** nd <= ids, nd + 1, nd), 1)
            lo = 0
            while lo < ids.shape[0]:                                              # ids ascend: runs of equal digit count are contiguous
                d = int(nd[lo])
                hi = lo + int(np.searchsorted(nd[lo:], d, side="right"))
                m = hi - lo
                row = np.empty((m, pre.shape[0] + d + 1 + L + 3 + L + 1), dtype=np.uint8)
                c = pre.shape[0]
                row[:, :c] = pre
                for k in range(d):
                    row[:, c + k] = (ids[lo:hi] // 10 ** (d - 1 - k)) % 10 + 48
                c += d
                row[:, c] = 10
                row[:, c + 1:c + 1 + L] = alpha[reads[s0 + lo:s0 + hi]]
                c += 1 + L
                row[:, c:c + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
                row[:, c + 3:c + 3 + L] = ord("I")
                row[:, c + 3 + L] = 10
                fh.write(row.tobytes())
                lo = hi


def apply_errors(frag: np.ndarray, read_len: int, rng, sub_rate: float, indel_rate: float) -> np.ndarray:
    """Rows of `frag` ([m, span], span > read_len: slack for deletions) -> reads [m, read_len] with substitutions (rate per base) and
    single-base insertions / deletions (indel_rate per base, half each; neighbouring events make longer gaps).  Events are drawn sparsely
    (their expected number of random places), so the cost is a few passes over the matrix whatever the rates."""
    m, span = frag.shape
    frag = np.ascontiguousarray(frag, dtype=np.uint8)
    flat = frag.reshape(-1)
    k = int(rng.binomial(flat.shape[0], sub_rate))
    at = rng.integers(0, flat.shape[0], size=k)
    flat[at] = (flat[at] + rng.integers(1, 4, size=k, dtype=np.uint8)) & 3
    if indel_rate <= 0:
        return np.ascontiguousarray(frag[:, :read_len])
    n_ev = rng.binomial(read_len, indel_rate, size=m)
    idx = np.broadcast_to(np.arange(read_len, dtype=np.int16)[None, :], (m, read_len)).copy()       # source column of every read base
    ar = np.arange(read_len, dtype=np.int16)[None, :]
    ins_r, ins_c = [], []
    for e in range(int(n_ev.max()) if m else 0):
        rows = np.nonzero(n_ev > e)[0]
        col = rng.integers(1, read_len - 1, size=rows.shape[0]).astype(np.int16)
        is_del = rng.random(rows.shape[0]) < 0.5
        # deletion in front of read column c: the tail comes from one base further; insertion at c: the tail from one base nearer
        idx[rows] += np.where(ar >= col[:, None], np.where(is_del, 1, 0)[:, None], 0).astype(np.int16)
        idx[rows] -= np.where(ar > col[:, None], np.where(is_del, 0, 1)[:, None], 0).astype(np.int16)
        ins_r.append(rows[~is_del]); ins_c.append(col[~is_del])
    np.clip(idx, 0, span - 1, out=idx)
    out = np.take_along_axis(frag, idx.astype(np.int64), axis=1)
    if ins_r:
        r, c = np.concatenate(ins_r), np.concatenate(ins_c).astype(np.int64)
        out[r, c] = rng.integers(0, 4, size=r.shape[0], dtype=np.uint8)
    return out


def make_pairs_chunk(genome: np.ndarray, m: int, read_len: int, rng, sub_rate: float, indel_rate: float = 0.0, ins_lo: int = 300, ins_hi: int = 500):
    """m read pairs (FR orientation, insert ins_lo..ins_hi + (read_len - 150)) of the forward strand of `genome`: (r1, r2) codes [m, read_len]."""
    slack = 0 if indel_rate <= 0 else max(16, int(read_len * indel_rate * 4) + 8)
    span = read_len + slack
    extra = read_len - 150
    pos = rng.integers(0, genome.shape[0] - ins_hi - extra - 2 * span - 8, size=m)
    ins = rng.integers(ins_lo, ins_hi, size=m) + extra
    win = np.lib.stride_tricks.sliding_window_view(genome, span)                 # row p = genome[p : p + span]: a gather of contiguous rows
    r1 = apply_errors(win[pos], read_len, rng, sub_rate, indel_rate)
    # the mate: reverse complement of the fragment's far end; its slack extends into the fragment
    end = pos + ins                                                              # one past the fragment's last base
    r2 = apply_errors(3 - win[end - span][:, ::-1], read_len, rng, sub_rate, indel_rate)
    return r1, r2


def repeat_dense_genome(l_pac, seed=77):
    """A genome at human-like repeat density (round 6; SURVEY 8(d) realises its configs on a 98 %-unique text): 42 % of the bases are copies of 300-bp
    interspersed repeat families -- 30 % old ones at 12 % divergence from their consensus (24 families, thousands of copies each: they share few 19-mers)
    and 12 % young ones at 2 % (2 families of ten thousand copies and more: 30-mers shared by thousands of them -> hit lists beyond max_occ) --, eight
    tandem satellites of a 171-bp monomer (300 copies each, 2 % divergence), two dozen 3-kb exact duplications and a dozen homopolymer runs."""
    return synth.make_genome(l_pac, seed=seed, repeat_frac=0.30, repeat_len=300, n_families=24, divergence=0.12, n_dups=24, dup_len=3000, poly_runs=12, satellites=8,
                             young_frac=0.12, young_families=2, young_divergence=0.02)
