"""Shared helpers for the test-suite (index building through our host tool, FASTQ parsing)."""
import os
import shutil
import subprocess
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
_TAB = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _TAB[_c] = _i
    _TAB[_c + 32] = _i

_CACHE = {}


def read_fastq_codes(path):
    """FASTQ -> (codes concatenated, offsets[n+1])"""
    seqs = []
    with open(path, "rb") as fh:
        lines = fh.read().split(b"\n")
    for i in range(1, len(lines), 4):
        if lines[i - 1].startswith(b"@"):
            seqs.append(_TAB[np.frombuffer(lines[i], dtype=np.uint8)])
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([s.shape[0] for s in seqs])
    return np.concatenate(seqs), off


def build_index(fasta, bits=12, threads=4):
    """Runs our meme-index on a copy of `fasta` in a temp dir; returns the index prefix (cached)."""
    key = (os.path.abspath(fasta), bits)
    if key in _CACHE:
        return _CACHE[key]
    d = tempfile.mkdtemp(prefix="memeidx_")
    dst = os.path.join(d, os.path.basename(fasta))
    shutil.copy(fasta, dst)
    subprocess.run([os.path.join(REPO, "bwa-meme_amd", "meme-index"), "build", dst, "-b", str(bits), "-t",
                    str(threads)], check=True, capture_output=True)
    _CACHE[key] = dst
    return dst


def chain_golden_workload():
    """The genome and reads tests/golden/chain_golden.npz was generated from (same seeds as tests/golden/make_chain_golden.py):
    returns (genome, list of reads as uint8 code arrays of their own lengths)."""
    from pymeme import synth
    g = synth.make_genome(300_000, seed=201, repeat_frac=0.15, repeat_len=250, n_families=5, divergence=0.02, n_dups=8, dup_len=1200, poly_runs=4)
    r1, _, _ = synth.make_reads(g, 2000, 150, seed=202, n_frac=0.03, exact_frac=0.2)
    r2, _, _ = synth.make_reads(g, 500, 250, seed=203, sub_rate=0.05, indel_rate=0.0075, n_frac=0.02)
    r3, _, _ = synth.make_reads(g, 300, 60, seed=204, sub_rate=0.02)
    reads, k = [], 0
    for rs in (r1, r2, r3):
        for r in rs:
            L = len(r) if rs is not r3 else 15 + k % 46
            reads.append(np.ascontiguousarray(r[:L], dtype=np.uint8))
            k += 1
    return g, reads
