#!/usr/bin/env python3
"""Generates tests/golden/sam_golden.npz: the SAM text the COMPILED REFERENCE's mem_aln2sam (src/bwamem.cpp:2174-2312, through oracle/_ref/libstage_ref.so
ref_aln2sam) writes for the records of tests/common.py sam_workload(), with hard and soft clipping, without and with a read group.  Runs in the build
container (no GPU).  Data only: the reference's outputs; the inputs are regenerated from seeds by the tests."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))
import oracle_py as O  # noqa: E402
import ref_py  # noqa: E402
from common import sam_workload  # noqa: E402

VARIANTS = ((0, b""), (1, b"grp1"))


def main(out):
    recs, blob, names, reads, quals, contigs = sam_workload()
    cb, co = O.contig_table(contigs)
    data = {}
    for softclip, rg in VARIANTS:
        texts = [ref_py.aln2sam(recs[k], blob, names[k], reads[k], quals[k], cb, co, softclip, rg) for k in range(recs.shape[0])]
        off = np.zeros(len(texts) + 1, np.int64)
        off[1:] = np.cumsum([len(t) for t in texts])
        data["text_%d" % softclip] = np.frombuffer(b"".join(texts), dtype=np.uint8)
        data["off_%d" % softclip] = off
    np.savez_compressed(out, **data)
    print("records", recs.shape[0], "bytes", {k: int(v.shape[0]) for k, v in data.items() if k.startswith("text")})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "sam_golden.npz"))
