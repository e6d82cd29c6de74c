"""Mate rescue whole on the device (round 6): the posing step -- mem_sam_pe_batch_pre + mem_matesw_batch_pre, reference
src/bwamem_pair.cpp:660-716, 1060-1223 -- and the Smith-Waterman jobs it poses (mem_sam_pe_batch, :719-818) through meme_matesw_batch_host,
against tests/golden/matesw_golden.npz (the compiled reference's own posing function and its AVX-512 kswv kernels over the same alignment
records) and, on other workloads and options, against the oracle's restatements (orc_matesw_pose, orc_kswv_batch: both pinned on the reference)."""
import os

import numpy as np
import pytest

import oracle_py as O
from common import GOLDEN, matesw_pose_workload
from pymeme import hipapi

pytestmark = pytest.mark.gpu


def _stage(ctx, genome, reads, read_off):
    """an index of the genome and the reads resident on the ctx (the stage names reads of the batch a seeding call left there)"""
    import torch
    l_pac = genome.shape[0]
    n = 2 * l_pac
    text = hipapi.fwd_rc_text(genome)
    d_text, d_sa = hipapi.build_sa_device(ctx, text)
    d_pos5 = hipapi.pos5_from_sa_torch(ctx, d_sa, n)
    del d_sa
    d_pac, d_ent = hipapi.stage_entries_torch(ctx, n, d_text, d_pos5)
    d_l2, n_l2, d_l1, n_l1 = hipapi.train_prmi_device(ctx, d_ent, n, 12)
    keep = (d_pac, d_ent) + hipapi.attach_index_torch(ctx, n, d_pac, d_ent, d_l2, n_l2, d_l1, n_l1)
    ctx.seed_batch_host(reads, read_off)
    return keep


def _check(ctx, W, gold=None, **kw):
    contigs = [(int(o), int(l), 0) for o, l in zip(W["contig_off"], W["contig_len"])]
    R = ctx.matesw_batch_host(W["regs"], W["reg_off"], W["pes"], contigs, int(W["l_pac"]), **kw)
    n = W["read_len"].shape[0]
    okw = {k: v for k, v in kw.items() if k in ("a", "pen_unpaired", "max_matesw", "min_seed_len")}
    for b, first in enumerate(range(0, n, 512)):
        count = min(512, n - first)
        gar, jobs = O.matesw_pose(W["regs"], W["reg_off"], first, count, W["read_len"], W["pes"], int(W["l_pac"]), W["contig_off"], W["contig_len"], **okw)
        g0, g1 = R["gar_off"][b], R["gar_off"][b + 1]
        j0, j1 = R["job_off"][b], R["job_off"][b + 1]
        assert np.array_equal(R["gar"][g0:g1], gar), ("gar of batch", b)
        dj = R["jobs"][j0:j1]
        assert dj.shape[0] == jobs.shape[0]
        assert np.array_equal(dj["len1"], jobs["len1"]) and np.array_equal(dj["len2"], jobs["len2"]) and np.array_equal(dj["xtra"], jobs["xtra"])
        if jobs.shape[0]:
            kj, ref, qer = O.matesw_job_seqs(jobs, W["text"], W["reads"], W["read_off"])
            want, _ = O.kswv_batch(kj, ref, qer, a=kw.get("a", 1), b=kw.get("b", 4))
            got = R["res"][j0:j1]
            for f in want.dtype.names:
                assert np.array_equal(got[f], want[f]), ("kswr field", f, "batch", b)
        if gold is not None:
            tag, z = gold
            assert np.array_equal(R["gar"][g0:g1], z[tag + "_gar"][z[tag + "_gar_off"][b]:z[tag + "_gar_off"][b + 1]])
            zr = z[tag + "_kswr"][z[tag + "_job_off"][b]:z[tag + "_job_off"][b + 1]]
            got = R["res"][j0:j1]
            for f in got.dtype.names:
                assert np.array_equal(got[f], zr[f]), ("golden kswr", f, b)
    return R


@pytest.mark.parametrize("tag", ["a", "b"])
def test_mate_rescue_on_the_device_equals_the_reference_golden(tag):
    z = np.load(os.path.join(GOLDEN, "matesw_golden.npz"))
    g = z[tag + "_genome"]
    W = {k: z[tag + "_" + k] for k in ("contig_off", "contig_len", "reads", "read_off", "read_len", "regs", "reg_off", "pes")}
    W["l_pac"] = int(z[tag + "_l_pac"]); W["genome"] = g; W["text"] = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
    ctx = hipapi.Context(0)
    try:
        keep = _stage(ctx, g, W["reads"], W["read_off"])
        R = _check(ctx, W, gold=(tag, z))
        assert R["jobs"].shape[0] == z[tag + "_len1"].shape[0] > 500
        # the same from a second ctx of the GPU that reads the first one's resident batch (how the binding runs the stage beside the CIGAR stage)
        other = hipapi.Context(0)
        other.attach_index(ctx.describe_index())
        contigs = [(int(o), int(l), 0) for o, l in zip(W["contig_off"], W["contig_len"])]
        R2 = other.matesw_batch_host(W["regs"], W["reg_off"], W["pes"], contigs, int(W["l_pac"]), reads_of=ctx)
        assert np.array_equal(R2["gar"], R["gar"]) and hipapi.records_equal(R2["res"], R["res"])
        other.close()
        # a sub-range of the resident batch (the binding runs a chunk's SAM phase in two halves): reads [512, n) as a call of its own = the whole call's batches 1..
        lo = 512
        R3 = ctx.matesw_batch_host(W["regs"][W["reg_off"][lo]:], W["reg_off"][lo:] - W["reg_off"][lo], W["pes"], contigs, int(W["l_pac"]), first_read=lo)
        assert np.array_equal(R3["gar"], R["gar"][R["gar_off"][1]:]) and hipapi.records_equal(R3["res"], R["res"][R["job_off"][1]:])
        assert np.array_equal(R3["job_off"], R["job_off"][1:] - R["job_off"][1])
        del keep
    finally:
        ctx.close()


@pytest.mark.parametrize("seed,pes,kw", [(321, [(10, 2000, 0)] * 4, {}), (322, [(0, 0, 1)] * 4, {}), (323, [(50, 500, 0), (120, 680, 0), (100, 900, 0), (0, 0, 1)], dict(max_matesw=2, pen_unpaired=40)),
                                         (324, None, dict(min_seed_len=40, a=2, b=5)), (325, [(0, 0, 1), (200, 260, 0), (0, 0, 1), (150, 151, 0)], dict(batch_reads=64))])
def test_mate_rescue_on_the_device_equals_the_oracle(seed, pes, kw):
    W = matesw_pose_workload(seed=seed, pes=pes, n_pairs=700)
    ctx = hipapi.Context(0)
    try:
        keep = _stage(ctx, W["genome"], W["reads"], W["read_off"])
        if kw.get("batch_reads", 512) != 512:
            # other worker-batch sizes only move the batch boundaries: per-batch job numbering restarts there
            contigs = [(int(o), int(l), 0) for o, l in zip(W["contig_off"], W["contig_len"])]
            R = ctx.matesw_batch_host(W["regs"], W["reg_off"], W["pes"], contigs, int(W["l_pac"]), **kw)
            n, br = W["read_len"].shape[0], kw["batch_reads"]
            for b, first in enumerate(range(0, n, br)):
                gar, jobs = O.matesw_pose(W["regs"], W["reg_off"], first, min(br, n - first), W["read_len"], W["pes"], int(W["l_pac"]), W["contig_off"], W["contig_len"])
                assert np.array_equal(R["gar"][R["gar_off"][b]:R["gar_off"][b + 1]], gar), b
                assert R["job_off"][b + 1] - R["job_off"][b] == jobs.shape[0]
        else:
            _check(ctx, W, **kw)
        # reads beyond the batch on the ctx, or an odd first read, are refused
        n = W["read_len"].shape[0]
        with pytest.raises(hipapi.MemeError):
            ctx.matesw_batch_host(W["regs"][:0], np.zeros(n + 3, np.int64), W["pes"], [(0, int(W["l_pac"]), 0)], int(W["l_pac"]))
        with pytest.raises(hipapi.MemeError):
            ctx.matesw_batch_host(W["regs"][:0], np.zeros(3, np.int64), W["pes"], [(0, int(W["l_pac"]), 0)], int(W["l_pac"]), first_read=1)
        del keep
    finally:
        ctx.close()
