"""Deterministic synthetic genomes and reads (numpy ``default_rng(seed)``).

There is no network on the build or GPU boxes, so every BASELINE.json config is realised with
synthetic data of the stated shape:

* genome = uniform random ACGT + interspersed repeat families (diverged copies of a few short
  elements) + a handful of long exact duplications, so that seeding meets unique loci, moderately
  repetitive seeds and high-copy seeds (the three regimes of the reference's round 1/2/3 seeding);
* reads  = fixed-length substrings of either strand with substitutions, small indels and a few
  ``N`` bases.

Bases are handled as uint8 codes 0..3 = A,C,G,T and 4 = N -- the same code space the reference's
``nst_nt4_table`` produces (reference src/bwamem.cpp:1277-1279).
"""
from __future__ import annotations

import numpy as np

_ALPHA = np.frombuffer(b"ACGTN", dtype=np.uint8)


def make_genome(n_bases: int, seed: int = 11, repeat_frac: float = 0.02, repeat_len: int = 300,
                n_families: int = 20, divergence: float = 0.03, n_dups: int = 8,
                dup_len: int = 2000, poly_runs: int = 4, satellites: int = 0, sat_unit: int = 171,
                sat_copies: int = 300, sat_divergence: float = 0.02, young_frac: float = 0.0,
                young_families: int = 6, young_divergence: float = 0.03) -> np.ndarray:
    """Return a uint8 array of codes 0..3 of length ``n_bases``.  ``satellites`` > 0 (the repeat-dense workloads) plants that many
    tandem arrays -- ``sat_copies`` copies of a ``sat_unit``-base monomer, each copy diverged by ``sat_divergence`` -- AFTER everything
    else, so that the genomes of existing seeds do not change."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=n_bases, dtype=np.uint8)
    if n_bases < 4 * repeat_len:
        return g
    # repeat families: each family is one consensus; copies are planted with per-base divergence
    fams = rng.integers(0, 4, size=(n_families, repeat_len), dtype=np.uint8)
    n_copies = int(n_bases * repeat_frac / repeat_len)
    starts = rng.integers(0, n_bases - repeat_len, size=n_copies)
    fam_id = rng.integers(0, n_families, size=n_copies)
    for s, f in zip(starts, fam_id):
        copy = fams[f].copy()
        mut = rng.random(repeat_len) < divergence
        copy[mut] = (copy[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
        g[s:s + repeat_len] = copy
    # long exact duplications (forces long LCPs between suffixes)
    for _ in range(n_dups):
        if n_bases < 4 * dup_len:
            break
        a = int(rng.integers(0, n_bases - dup_len))
        b = int(rng.integers(0, n_bases - dup_len))
        g[b:b + dup_len] = g[a:a + dup_len]
    # a few homopolymer / dinucleotide runs (stress the T-padding rule of the suffix array order)
    for k in range(poly_runs):
        L = int(rng.integers(20, 60))
        s = int(rng.integers(0, n_bases - L))
        g[s:s + L] = k & 3
    # satellites: tandem arrays of a short monomer (alpha-satellite-like: hundreds of near-identical copies side by side)
    for _ in range(satellites):
        L = sat_unit * sat_copies
        if n_bases < 4 * L:
            break
        mono = rng.integers(0, 4, size=sat_unit, dtype=np.uint8)
        arr = np.tile(mono, sat_copies)
        mut = rng.random(L) < sat_divergence
        arr[mut] = (arr[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
        s = int(rng.integers(0, n_bases - L))
        g[s:s + L] = arr
    # young repeat families (the repeat-dense workloads): copies a few per cent from their consensus, so that 19-mers ARE shared between
    # hundreds of copies -- seeds with hit lists beyond max_occ, reads whose SMEM slots overflow (planted last: existing seeds unchanged)
    if young_frac > 0:
        yf = rng.integers(0, 4, size=(young_families, repeat_len), dtype=np.uint8)
        n_copies = int(n_bases * young_frac / repeat_len)
        starts = rng.integers(0, n_bases - repeat_len, size=n_copies)
        fam_id = rng.integers(0, young_families, size=n_copies)
        mut = rng.random((n_copies, repeat_len)) < young_divergence
        shift = rng.integers(1, 4, size=(n_copies, repeat_len), dtype=np.uint8)
        copies = np.where(mut, (yf[fam_id] + shift) & 3, yf[fam_id]).astype(np.uint8)
        for k in range(n_copies):
            g[starts[k]:starts[k] + repeat_len] = copies[k]
    return g


def revcomp(codes: np.ndarray) -> np.ndarray:
    out = codes[::-1].copy()
    m = out < 4
    out[m] = 3 - out[m]
    return out


def make_reads(genome: np.ndarray, n_reads: int, read_len: int = 150, seed: int = 12,
               sub_rate: float = 0.01, indel_rate: float = 0.0015, n_frac: float = 0.02,
               exact_frac: float = 0.0):
    """Sample reads.  Returns (codes[n_reads, read_len] uint8, origin_pos int64, strand uint8).

    Substitutions are applied vectorised; indels (rare) read-by-read on the affected reads only.
    ``exact_frac`` of the reads are left error-free (exercises the whole-read-match path,
    reference src/LearnedIndex_seeding.cpp:1870-1880).
    """
    rng = np.random.default_rng(seed)
    n = genome.shape[0]
    span = read_len + 8  # slack so deletions can be compensated
    pos = rng.integers(0, n - span, size=n_reads)
    idx = pos[:, None] + np.arange(span)[None, :]
    frag = genome[idx]  # [n_reads, span]
    strand = rng.integers(0, 2, size=n_reads).astype(np.uint8)
    exact = rng.random(n_reads) < exact_frac
    # substitutions
    sub = rng.random((n_reads, span)) < sub_rate
    sub[exact] = False
    delta = rng.integers(1, 4, size=(n_reads, span), dtype=np.uint8)
    frag = np.where(sub, (frag + delta) & 3, frag).astype(np.uint8)
    reads = frag[:, :read_len].copy()
    # indels on a subset
    has_indel = (rng.random(n_reads) < indel_rate * read_len) & ~exact
    for r in np.nonzero(has_indel)[0]:
        row = list(frag[r])
        p = int(rng.integers(5, read_len - 5))
        L = int(rng.integers(1, 4))
        if rng.random() < 0.5:
            del row[p:p + L]
        else:
            row[p:p] = list(rng.integers(0, 4, size=L))
        reads[r] = np.asarray(row[:read_len], dtype=np.uint8)
    # reverse strand
    rc_rows = np.nonzero(strand)[0]
    if rc_rows.size:
        reads[rc_rows] = 3 - reads[rc_rows][:, ::-1]
    # ambiguous bases
    with_n = (rng.random(n_reads) < n_frac) & ~exact
    rows = np.nonzero(with_n)[0]
    cols = rng.integers(0, read_len, size=rows.size)
    reads[rows, cols] = 4
    return reads, pos.astype(np.int64), strand


def write_fasta(path: str, genome: np.ndarray, name: str = "chrS", width: int = 80,
                contigs: int = 1) -> None:
    """Write the genome as ``contigs`` equally sized FASTA records."""
    n = genome.shape[0]
    bounds = np.linspace(0, n, contigs + 1).astype(np.int64)
    with open(path, "wb") as fh:
        for c in range(contigs):
            seq = _ALPHA[genome[bounds[c]:bounds[c + 1]]].tobytes()
            fh.write(b">%s%d\n" % (name.encode(), c + 1))
            for i in range(0, len(seq), width):
                fh.write(seq[i:i + width])
                fh.write(b"\n")


def write_fastq(path: str, reads: np.ndarray, prefix: str = "r") -> None:
    qual = b"I" * reads.shape[1]
    with open(path, "wb") as fh:
        for i in range(reads.shape[0]):
            fh.write(b"@%s%d\n" % (prefix.encode(), i))
            fh.write(_ALPHA[reads[i]].tobytes())
            fh.write(b"\n+\n")
            fh.write(qual)
            fh.write(b"\n")


def read_fasta_codes(path: str) -> np.ndarray:
    """FASTA -> codes 0..4 (all records concatenated)."""
    tab = np.full(256, 4, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        tab[ch] = i
        tab[ch + 32] = i
    chunks = []
    with open(path, "rb") as fh:
        for line in fh:
            if line.startswith(b">"):
                continue
            chunks.append(np.frombuffer(line.rstrip(), dtype=np.uint8))
    return tab[np.concatenate(chunks)]
