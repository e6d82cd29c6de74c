#!/usr/bin/env python3
"""Generates tests/golden/ext_golden.npz: the alignment records the COMPILED REFERENCE's own mem_chain2aln_across_reads_V2 (with its
own BandedPairWiseSW kernels; through oracle/_ref/libstage_ref.so, oracle/ref_stage_shim.cpp) makes of the reads and chains of
tests/golden/chain_golden.npz (2 800 reads of a repeat-rich genome: 150 bp, 250 bp with 5 % errors + indels, 15-60 bp; chains made by
the reference's mem_chain_Learned + mem_chain_flt), plus 300 reads with an 80-95-base gap next to the seed (band retry).
Runs in the build container (no GPU).  Data only: the reference's outputs; the inputs are regenerated from seeds by the tests."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))
import oracle_py as O  # noqa: E402
import ref_py  # noqa: E402
from common import ext_golden_inputs  # noqa: E402


def main(out):
    I = ext_golden_inputs()
    regs = ref_py.extend_reads(I["reads"], I["read_off"], I["chain_off"], I["chains"], I["seed_off"], I["seeds"], I["frac_rep"], I["text"], I["l_pac"],
                               I["contig_off"], I["contig_len"], np.zeros(I["contig_off"].shape[0], np.uint8), O.default_ext_opt())
    cols = np.stack([regs[f].astype(np.int64) for f in O.ALNREG_FIELDS], 1)
    np.savez_compressed(out, regs=cols, frac_rep_bits=regs["frac_rep"].view(np.uint32), reg_off=I["seed_off"])
    purged = int(((regs["qb"] == -1) & (regs["qe"] == -1)).sum())
    print("reads", I["read_off"].shape[0] - 1, "records", regs.shape[0], "purged", purged, "with the doubled band", int((regs["w"] > 100).sum()))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ext_golden.npz"))
