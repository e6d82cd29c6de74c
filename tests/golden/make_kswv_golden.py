#!/usr/bin/env python3
"""Generates tests/golden/kswv_golden.npz: the kswr_t records (score, te, qe, score2, te2, tb, qb) the COMPILED REFERENCE's mate-rescue batch
(sort_classify + mem_sam_pe_batch with the AVX-512 kswv kernels, src/bwamem.cpp:1798-1825, src/bwamem_pair.cpp:719-818, src/kswv.cpp, through
oracle/_ref/libstage_ref.so) gives for the jobs of tests/common.py kswv_workload(), under three sets of scoring parameters.  Runs in the
build container (no GPU).  Data only: the reference's outputs; the inputs are regenerated from seeds by the tests."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))
import ref_py  # noqa: E402
from common import KSWV_GOLDEN_SETS, kswv_workload  # noqa: E402


def main(out):
    data = {}
    for name, kw, pen in KSWV_GOLDEN_SETS:
        jobs, ref, qer = kswv_workload(**kw)
        r = ref_py.kswv_batch(jobs, ref, qer, **pen)
        data[name] = r.view(np.int32).reshape(-1, 7)
        print(name, "jobs", r.shape[0], "int8 class", int(((jobs["xtra"] & 0x10000) != 0).sum()), "with start", int((r["tb"] >= 0).sum()), "with a second-best score",
              int((r["score2"] > 0).sum()))
    np.savez_compressed(out, **data)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "kswv_golden.npz"))
