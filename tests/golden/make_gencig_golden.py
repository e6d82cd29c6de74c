#!/usr/bin/env python3
"""Generates tests/golden/gencig_golden.npz: score, CIGAR, NM and MD string the COMPILED REFERENCE's bwa_gen_cigar2 (src/bwa.cpp:274-362, through
oracle/_ref/libstage_ref.so ref_gen_cigar2) gives for the calls of tests/common.py gencig_workload().  Runs in the build container (no GPU).
Data only: the reference's outputs; the inputs are regenerated from seeds by the tests."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))
import ref_py  # noqa: E402
from common import gencig_workload  # noqa: E402


def main(out):
    g, reads, calls = gencig_workload()
    score, ncig, nm, cigs, mds = [], [], [], [], []
    for J in calls:
        q = reads[int(J["read"])][int(J["qb"]):int(J["qb"]) + int(J["qlen"])]
        r = ref_py.gen_cigar2(g, q, int(J["rb"]), int(J["rb"]) + int(J["tlen"]), int(J["w_"]))
        assert r is not None
        score.append(r[0]); ncig.append(r[1].shape[0]); nm.append(r[2]); cigs.append(r[1]); mds.append(r[3] + b"\0")
    np.savez_compressed(out, score=np.array(score, np.int32), n_cigar=np.array(ncig, np.int32), nm=np.array(nm, np.int32), cigars=np.concatenate(cigs),
                        md=np.frombuffer(b"".join(mds), dtype=np.uint8))
    print("calls", len(score), "operations", int(np.sum(ncig)), "gap-free shortcut", int(np.sum((calls["qlen"] == calls["tlen"]) & (calls["w_"] == 0))),
          "reverse strand", int(np.sum(calls["rb"] >= g.shape[0])), "MD bytes", sum(len(m) for m in mds))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "gencig_golden.npz"))
