#!/usr/bin/env python3
"""Generates tests/golden/gcig_golden.npz: score and CIGAR the COMPILED REFERENCE's ksw_global2 (src/ksw.cpp:560-670, through
oracle/_ref/libstage_ref.so) gives for the jobs of tests/common.py gcig_workload().  Runs in the build container (no GPU).
Data only: the reference's outputs; the inputs are regenerated from seeds by the tests."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))
import ref_py  # noqa: E402
from common import gcig_workload  # noqa: E402


def main(out):
    _, _, jobs, seqs = gcig_workload()
    scores, ncig, cigs = [], [], []
    for J, (q, t) in zip(jobs, seqs):
        sc, cg = ref_py.ksw_global2(q, t, int(J["w"]))
        scores.append(sc); ncig.append(cg.shape[0]); cigs.append(cg)
    np.savez_compressed(out, score=np.array(scores, np.int32), n_cigar=np.array(ncig, np.int32), cigars=np.concatenate(cigs))
    print("jobs", len(scores), "operations", int(np.sum(ncig)), "with gaps", int(np.sum(np.array(ncig) > 1)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "gcig_golden.npz"))
