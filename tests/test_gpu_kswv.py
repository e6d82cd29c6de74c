"""The mate-rescue Smith-Waterman kernel (meme_kswv_batch_host) against the compiled reference's fixture and the oracle."""
import os

import numpy as np
import pytest

import oracle_py as O
from common import GOLDEN, KSWV_EDGE_WANT, KSWV_GOLDEN_SETS, kswv_edge_jobs, kswv_workload
from pymeme import hipapi

pytestmark = pytest.mark.gpu


def _opt(a=1, b=4, o_del=6, e_del=1, o_ins=6, e_ins=1):
    o = hipapi.default_bsw_opt()
    o.a, o.b, o.o_del, o.e_del, o.o_ins, o.e_ins = a, b, o_del, e_del, o_ins, e_ins
    return o


def test_kswv_kernel_equals_reference_golden():
    """All seven kswr_t fields of 5 500 jobs == what sort_classify + mem_sam_pe_batch of the compiled reference (AVX-512 kswv kernels) gave
    (tests/golden/kswv_golden.npz): int8 and int16 classes, second-best scores, reverse passes, N, low-complexity sequence, other penalties."""
    G = np.load(os.path.join(GOLDEN, "kswv_golden.npz"))
    ctx = hipapi.Context(0)
    try:
        for name, kw, pen in KSWV_GOLDEN_SETS:
            jobs, ref, qer = kswv_workload(**kw)
            got, ms = ctx.kswv_batch_host(jobs.view(hipapi.KSWV_JOB), ref, qer, _opt(**pen))
            got = got.view(np.int32).reshape(-1, 7)
            bad = np.nonzero((got != G[name]).any(axis=1))[0]
            assert bad.size == 0, (name, int(bad.size), int(bad[0]), jobs[bad[0]].tolist(), got[bad[0]].tolist(), G[name][bad[0]].tolist())
            assert ms > 0
    finally:
        ctx.close()


def test_kswv_kernel_equals_oracle_on_fresh_jobs_and_edges():
    """Fresh seeds and penalties against orc_kswv_batch; plus edge jobs: a one-base window, a window shorter than the read, a query of one
    base, jobs without KSW_XSTART / without KSW_XSUBO, a caller-set KSW_XSTOP, an empty batch; malformed jobs are refused."""
    ctx = hipapi.Context(0)
    try:
        for kw, pen in ((dict(n=3000, seed=33), {}), (dict(n=1500, seed=34, read_len=(19, 140)), dict(a=1, b=9, o_del=1, e_del=1, o_ins=1, e_ins=1)),
                        (dict(n=1500, seed=35, read_len=(240, 500), a=3), dict(a=3, b=5, o_del=7, e_del=2, o_ins=3, e_ins=3))):
            jobs, ref, qer = kswv_workload(**kw)
            want = O.kswv_batch(jobs, ref, qer, **pen)[0].view(np.int32).reshape(-1, 7)
            got = ctx.kswv_batch_host(jobs.view(hipapi.KSWV_JOB), ref, qer, _opt(**pen))[0].view(np.int32).reshape(-1, 7)
            bad = np.nonzero((got != want).any(axis=1))[0]
            assert bad.size == 0, (kw, int(bad.size), int(bad[0]), jobs[bad[0]].tolist(), got[bad[0]].tolist(), want[bad[0]].tolist())
        jobs, rb, qb = kswv_edge_jobs()
        want = O.kswv_batch(jobs, rb, qb)[0].view(np.int32).reshape(-1, 7)
        got = ctx.kswv_batch_host(jobs.view(hipapi.KSWV_JOB), rb, qb)[0].view(np.int32).reshape(-1, 7)
        assert np.array_equal(got, want), (got.tolist(), want.tolist())
        assert np.array_equal(want, np.array(KSWV_EDGE_WANT, np.int32))     # (the compiled reference's records: tests/test_ref_live.py checks them live)
        g, q, X = rb, qb, int(jobs["xtra"][0])
        assert ctx.kswv_batch_host(np.zeros(0, hipapi.KSWV_JOB), g, q)[0].shape[0] == 0
        for bad_job in ((g.shape[0] - 100, 400, 0, 150, X), (0, 400, q.shape[0] - 50, 150, X), (0, 40000, 0, 150, X)):
            jb = np.zeros(1, hipapi.KSWV_JOB)
            jb[0] = (bad_job[0], bad_job[2], bad_job[1], bad_job[3], bad_job[4], 0)
            with pytest.raises(RuntimeError):
                ctx.kswv_batch_host(jb, g, q)
    finally:
        ctx.close()
