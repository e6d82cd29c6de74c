#!/usr/bin/env python3
"""Generates the committed golden vectors from the COMPILED REFERENCE (oracle/_ref).

Run in the build container only (needs oracle/_ref, built from /root/reference by
`make -f oracle/Makefile.ref`):

    python tests/golden/make_golden.py

Outputs (data only -- inputs and the reference's outputs):
  g1.fa            synthetic 48 kbp genome, 3 contigs (seed 101)
  g1_reads_<L>.fq  260 synthetic reads in four files: 150 bp / 250 bp 5 %-error / 60 bp N-rich / 25 bp
  g1_seeds_<L>.txt stdout of `learned_seeding_big_read <idx> g1_reads_<L>.fq 1000 1 4` (MODE=3), which the
                   script asserts to be identical to the MODE=1 build and to the FM-index harness
  bsw_golden.npz   2400 seed-extension tasks and the six outputs of the reference's
                   scalarBandedSWAWrapper (w=100 and w=200, end_bonus=5 and 0) plus getScores16
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))

import bsw_gen  # noqa: E402
import oracle_py as O  # noqa: E402
import ref_py as R  # noqa: E402
from pymeme import synth  # noqa: E402


def main():
    ref = os.path.join(REPO, "oracle", "_ref")
    assert os.path.isdir(ref), "build oracle/_ref first: make -f oracle/Makefile.ref"
    # ---- seeding ------------------------------------------------------------------------------
    g = synth.make_genome(48_000, seed=101, repeat_frac=0.12, repeat_len=200, n_families=3,
                          divergence=0.02, n_dups=4, dup_len=800, poly_runs=4)
    g[:5] = 0  # leading A-run => the fwd+rc text ends in a T-run (exercises the padding tie rule)
    synth.write_fasta(os.path.join(HERE, "g1.fa"), g, name="gold", contigs=3)
    r1, _, _ = synth.make_reads(g, 140, 150, seed=102, n_frac=0.1, exact_frac=0.3)
    r2, _, _ = synth.make_reads(g, 40, 250, seed=103, sub_rate=0.05, indel_rate=0.0075, n_frac=0.05)
    r3, _, _ = synth.make_reads(g, 40, 60, seed=104, sub_rate=0.02, n_frac=0.5)
    r4, _, _ = synth.make_reads(g, 40, 25, seed=105, sub_rate=0.0, n_frac=0.1)
    # one FASTQ per read length: the reference harness hangs on mixed-length input files
    fqs = []
    k = 0
    for rs in (r1, r2, r3, r4):
        fq = os.path.join(HERE, "g1_reads_%d.fq" % rs.shape[1])
        synth.write_fastq(fq, rs, prefix="g%d_" % rs.shape[1])
        fqs.append(fq)
        k += rs.shape[0]
    tmp = tempfile.mkdtemp(prefix="golden_")
    try:
        fa = os.path.join(tmp, "g1.fa")
        shutil.copy(os.path.join(HERE, "g1.fa"), fa)
        run = lambda *a, **kw: subprocess.run(a, check=True, capture_output=True, **kw)
        run(os.path.join(ref, "bwa-meme_mode3"), "index", "-a", "meme", "-t", "4", fa)
        run(os.path.join(ref, "bwa-meme_mode3"), "index", fa)  # FM-index for the differential harness
        run(os.path.join(REPO, "bwa-meme_amd", "meme-index"), "train", fa, "-b", "12")
        nl = 0
        for fq in fqs:
            d3 = R.run_seed_dump(fa, fq, mode=3, timeout=120)
            d1 = R.run_seed_dump(fa, fq, mode=1, timeout=120)
            dfm = run(os.path.join(ref, "fmi_seeding"), fa, fq, "1000", "1", "4").stdout.decode()
            assert d3 == d1, "reference MODE=3 and MODE=1 seed dumps differ"
            assert d3 == dfm, "reference learned and FM-index seed dumps differ"
            open(fq.replace("_reads_", "_seeds_").replace(".fq", ".txt"), "w").write(d3)
            nl += d3.count("\n")
        print("seeding golden: %d reads, %d dump lines" % (k, nl))
    finally:
        shutil.rmtree(tmp)
    # ---- banded SW ------------------------------------------------------------------------------
    sets = [bsw_gen.make_pairs(800, seed=201), bsw_gen.make_pairs(800, seed=202, max_q=250, sub=0.06, indel=0.02),
            bsw_gen.make_pairs(800, seed=203, max_q=40)]
    pairs = np.concatenate([s[0] for s in sets])
    ro = qo = 0
    for s in sets:
        s[0]["idr"] += ro
        s[0]["idq"] += qo
        ro += s[1].shape[0]
        qo += s[2].shape[0]
    pairs = np.concatenate([s[0] for s in sets])
    refb = np.concatenate([s[1] for s in sets])
    qerb = np.concatenate([s[2] for s in sets])
    out = {"pairs": pairs, "ref": refb, "qer": qerb}
    for w in (100, 200):
        for eb in (5, 0):
            prm = O.default_bsw_params(end_bonus=eb)
            out["scalar_w%d_eb%d" % (w, eb)] = bsw_gen.outputs(R.bsw_run(0, pairs, refb, qerb, w, prm))
    out["simd16_w100_eb5"] = bsw_gen.outputs(R.bsw_run(16, pairs, refb, qerb, 100, O.default_bsw_params(5)))
    np.savez_compressed(os.path.join(HERE, "bsw_golden.npz"), **out)
    print("bsw golden: %d pairs" % pairs.shape[0])


if __name__ == "__main__":
    main()
