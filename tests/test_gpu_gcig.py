"""The CIGAR kernel (meme_global_batch_host = ksw_global2 under bwa_gen_cigar2), through the C ABI, against score and CIGAR of the
compiled reference (tests/golden/gcig_golden.npz) and against the oracle with other penalties."""
import os

import numpy as np
import pytest

import oracle_py as O
from common import GOLDEN, build_index, gcig_workload, gencig_workload
from pymeme import hipapi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[1, 0, 2], ids=["groups", "wavefront-per-job", "groups+two-columns-per-lane"])
def _gcig_groups(request, monkeypatch):
    """Round 6: jobs with bands of at most 16 / 32 columns run 4 / 2 to a wavefront (k_gcig_grp); every test of this file runs with that (the default) and with
    one wavefront per job (k_gcig alone) -- the same scores, operations, NM and MD either way."""
    monkeypatch.setenv("MEME_TUNING", "gcig_groups=%d" % request.param)


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
    # (the workload's bands are 1, 3, 10, 30, 100, ...: every third job gets one of 33-63 instead -- 67-127 band columns, the two-columns-per-lane kernel of round 6)
    jobs = jobs.copy()
    for k in range(0, jobs.shape[0], 3):
        need = abs(int(jobs["tlen"][k]) - int(jobs["qlen"][k]))
        jobs["w"][k] = max(need, 33 + (7 * k) % 31)
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


def _check_calls(res, cig, md, want):
    """device results of a gen_cigar batch against a list of (score, cigar, nm, md) tuples"""
    assert res.shape[0] == len(want)
    o, m = 0, 0
    for k, (sc, cg, nm, s) in enumerate(want):
        R = res[k]
        assert int(R["cigar_off"]) == o and int(R["md_off"]) == m, k                # packed in job order
        got_md = md[m:m + int(R["md_len"])].tobytes()
        assert md[m + int(R["md_len"])] == 0
        assert (int(R["score"]), int(R["nm"]), got_md) == (sc, nm, s) and np.array_equal(cig[o:o + int(R["n_cigar"])], cg), (k, R, sc, nm, s, got_md)
        o += int(R["n_cigar"]); m += int(R["md_len"]) + 1
    assert o == cig.shape[0] and m == md.shape[0]


def test_device_gen_cigar_equals_reference_golden(tmp_path):
    """meme_gen_cigar_batch_host (bwa_gen_cigar2 whole on the device: shortcut or band + DP + traceback, NM, MD) against what the compiled
    reference's function returned for the same calls (tests/golden/gencig_golden.npz)."""
    g, reads, calls = gencig_workload()
    G = np.load(os.path.join(GOLDEN, "gencig_golden.npz"))
    off = np.concatenate([[0], np.cumsum(G["n_cigar"])])
    mds = G["md"].tobytes().split(b"\0")
    want = [(int(G["score"][k]), G["cigars"][off[k]:off[k + 1]], int(G["nm"][k]), mds[k]) for k in range(calls.shape[0])]
    ctx = _ctx_with_reads(tmp_path, g, reads)
    try:
        res, cig, md, ms = ctx.gen_cigar_batch_host(calls)
        _check_calls(res, cig, md, want)
        # a call the reference's function rejects is refused, not computed: a target across the strand boundary, an empty query
        bad = calls[:3].copy()
        bad["rb"][1] = g.shape[0] - 10; bad["tlen"][1] = 40
        with pytest.raises(hipapi.MemeError, match="malformed"):
            ctx.gen_cigar_batch_host(bad)
        bad = calls[:3].copy()
        bad["qb"][2] = 5; bad["qlen"][2] = len(reads[int(bad["read"][2])])
        with pytest.raises(hipapi.MemeError, match="beyond the end of read"):
            ctx.gen_cigar_batch_host(bad)
        r0 = ctx.gen_cigar_batch_host(calls[:0])
        assert r0[0].shape[0] == 0 and r0[1].shape[0] == 0 and r0[2].shape[0] == 0
    finally:
        ctx.close()


def test_device_gen_cigar_equals_oracle_with_other_penalties(tmp_path):
    g, reads, calls = gencig_workload(n=700, seed=211)
    text = hipapi.fwd_rc_text(g)
    ctx = _ctx_with_reads(tmp_path, g, reads)
    try:
        for a, b, od, ed, oi, ei in ((2, 3, 4, 2, 7, 1), (1, 9, 1, 1, 1, 1), (3, 1, 5, 3, 2, 2)):
            opt = hipapi.BswOpt(od, ed, oi, ei, 100, 5, a, b)
            res, cig, md, _ = ctx.gen_cigar_batch_host(calls, opt)
            want = [O.gen_cigar2(text, g.shape[0], reads[int(J["read"])][int(J["qb"]):int(J["qb"]) + int(J["qlen"])], int(J["rb"]), int(J["rb"]) + int(J["tlen"]),
                                 int(J["w_"]), a, b, od, ed, oi, ei) for J in calls]
            _check_calls(res, cig, md, want)
    finally:
        ctx.close()
