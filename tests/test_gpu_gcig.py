"""The CIGAR kernel (meme_global_batch_host = ksw_global2 under bwa_gen_cigar2), through the C ABI, against score and CIGAR of the
compiled reference (tests/golden/gcig_golden.npz) and against the oracle with other penalties."""
import os

import numpy as np
import pytest

import oracle_py as O
from common import GOLDEN, build_index, gcig_workload
from pymeme import hipapi, synth

pytestmark = pytest.mark.gpu


def _ctx_with_reads(tmp_path, g, reads):
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, g, contigs=2)
    prefix = build_index(fa, bits=14)
    ctx = hipapi.Context(0)
    ctx.load_index_files(prefix)
    off = np.zeros(len(reads) + 1, np.int64)
    off[1:] = np.cumsum([len(r) for r in reads])
    ctx.seed_batch_host(np.concatenate(reads), off)            # stages the reads the jobs refer to
    return ctx


def test_device_cigars_equal_reference_golden(tmp_path):
    g, reads, jobs, _ = gcig_workload()
    G = np.load(os.path.join(GOLDEN, "gcig_golden.npz"))
    ctx = _ctx_with_reads(tmp_path, g, reads)
    try:
        res, cig, ms = ctx.global_batch_host(jobs)
    finally:
        ctx.close()
    assert np.array_equal(res["score"], G["score"])
    assert np.array_equal(res["n_cigar"], G["n_cigar"])
    assert np.array_equal(res["cigar_off"], np.concatenate([[0], np.cumsum(G["n_cigar"])])[:-1])
    assert np.array_equal(cig, G["cigars"])


def test_device_cigars_equal_oracle_with_other_penalties(tmp_path):
    g, reads, jobs, seqs = gcig_workload(n=800, seed=91)
    ctx = _ctx_with_reads(tmp_path, g, reads)
    try:
        for a, b, od, ed, oi, ei in ((2, 3, 4, 2, 7, 1), (1, 9, 1, 1, 1, 1)):
            opt = hipapi.BswOpt(od, ed, oi, ei, 100, 5, a, b)
            res, cig, _ = ctx.global_batch_host(jobs, opt)
            for k, (J, (q, t)) in enumerate(zip(jobs, seqs)):
                sc, cg = O.ksw_global2(q, t, int(J["w"]), a, b, od, ed, oi, ei)
                o0 = int(res["cigar_off"][k])
                assert sc == int(res["score"][k]) and np.array_equal(cg, cig[o0:o0 + int(res["n_cigar"][k])]), (k, a, b, od, ed, oi, ei)
        bad = jobs[:4].copy()
        bad["read"][2] = len(reads) + 5
        with pytest.raises(hipapi.MemeError, match="malformed"):
            ctx.global_batch_host(bad)
        # a band that does not reach the matrix's last cell, and a query span that runs past its read: refused, not computed on garbage
        narrow = jobs[:4].copy()
        narrow["tlen"][1] = narrow["qlen"][1] + 40
        narrow["w"][1] = 10
        with pytest.raises(hipapi.MemeError, match="malformed"):
            ctx.global_batch_host(narrow)
        long_q = jobs[:4].copy()
        long_q["qb"][3] = 5
        long_q["qlen"][3] = len(reads[int(long_q["read"][3])])
        long_q["w"][3] = 500
        with pytest.raises(hipapi.MemeError, match="beyond the end of read"):
            ctx.global_batch_host(long_q)
    finally:
        ctx.close()
