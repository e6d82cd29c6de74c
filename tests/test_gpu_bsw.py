"""Parity of the HIP banded-SW kernel (through the C ABI) against the golden reference outputs and the oracle."""
import os

import numpy as np
import pytest

import bsw_gen
import oracle_py as O
from common import GOLDEN
from pymeme import hipapi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[0, 1 << 30], ids=["lane-per-pair", "lanes-per-pair"])
def ctx(request):
    # big batches take the lane-per-pair kernel, small ones the lanes-per-pair kernel: force each for every test
    c = hipapi.Context(0)
    c.set_tuning("bsw_lane_min_pairs", request.param)
    yield c
    c.close()


def _opt(eb):
    return hipapi.default_bsw_opt(end_bonus=eb)


@pytest.mark.parametrize("w,eb", [(100, 5), (100, 0), (200, 5), (200, 0)])
def test_hip_bsw_equals_reference_golden(ctx, w, eb):
    z = np.load(os.path.join(GOLDEN, "bsw_golden.npz"))
    pairs = z["pairs"].astype(hipapi.SEQPAIR).copy()
    ctx.bsw_batch(pairs, z["ref"], z["qer"], w, _opt(eb))
    assert np.array_equal(bsw_gen.outputs(pairs), z["scalar_w%d_eb%d" % (w, eb)])


@pytest.mark.parametrize("kw", [dict(), dict(max_q=300, sub=0.06, indel=0.02), dict(max_q=40), dict(max_q=500, h0_max=400),
                                dict(max_q=150, sub=0.3, unrelated_frac=0.5), dict(max_q=1000, min_q=600, h0_max=50)])
def test_hip_bsw_equals_oracle_random(ctx, kw):
    pairs, ref, qer = bsw_gen.make_pairs(1500, seed=11, **kw)
    for w in (100, 200, 7):
        want = pairs.copy()
        O.bsw_batch(want, ref, qer, w, O.default_bsw_params(5), threads=0)
        got = pairs.copy()
        ctx.bsw_batch(got, ref, qer, w, _opt(5))
        assert np.array_equal(bsw_gen.outputs(got), bsw_gen.outputs(want)), (kw, w)


def test_hip_bsw_other_scoring(ctx):
    pairs, ref, qer = bsw_gen.make_pairs(1500, seed=5, max_q=200)
    o = hipapi.BswOpt(4, 2, 8, 1, 50, 7, 2, 5)
    po = O.OrcBswParams(4, 2, 8, 1, 50, 7, 2, 5)
    want = pairs.copy(); O.bsw_batch(want, ref, qer, 60, po)
    got = pairs.copy(); ctx.bsw_batch(got, ref, qer, 60, o)
    assert np.array_equal(bsw_gen.outputs(got), bsw_gen.outputs(want))


def test_hip_bsw_edge_cases(ctx):
    # empty query / empty target / single bases / all-N / zero h0
    pairs = np.zeros(6, dtype=hipapi.SEQPAIR)
    ref = np.array([0, 1, 2, 3, 4, 4, 4, 0, 0, 0, 0, 0], np.uint8)
    qer = np.array([0, 1, 2, 3, 4, 4, 4, 0, 0, 0, 0, 0], np.uint8)
    spec = [(0, 0, 4, 0, 10), (0, 0, 0, 4, 10), (0, 0, 1, 1, 1), (4, 4, 3, 3, 20), (7, 7, 5, 5, 0), (0, 7, 4, 5, 3)]
    for i, (idr, idq, l1, l2, h0) in enumerate(spec):
        pairs[i]["idr"], pairs[i]["idq"], pairs[i]["len1"], pairs[i]["len2"], pairs[i]["h0"] = idr, idq, l1, l2, h0
    want = pairs.copy(); O.bsw_batch(want, ref, qer, 100, O.default_bsw_params(5))
    got = pairs.copy(); ctx.bsw_batch(got, ref, qer, 100, _opt(5))
    assert np.array_equal(bsw_gen.outputs(got), bsw_gen.outputs(want))
    assert ctx.bsw_batch(np.zeros(0, hipapi.SEQPAIR), ref, qer, 100).shape[0] == 0


@pytest.mark.parametrize("q", [1, 30, 31, 62, 63, 94, 95, 126, 127, 158, 159, 222, 223, 318, 319, 600, 601])
def test_hip_bsw_length_class_boundaries(ctx, q):
    # the lane-per-pair kernel is launched per LDS size class (query <= 30 / 62 / 94 / 126 / 158 / 222 / 318 / 600); longer queries and scores
    # beyond 14 bits take the lanes-per-pair kernel: every boundary, with a wavefront that is not full
    pairs, ref, qer = bsw_gen.make_pairs(130, seed=100 + q, min_q=q, max_q=q, h0_max=60)
    want = pairs.copy()
    O.bsw_batch(want, ref, qer, 100, O.default_bsw_params(5), threads=0)
    got = pairs.copy()
    ctx.bsw_batch(got, ref, qer, 100, _opt(5))
    assert np.array_equal(bsw_gen.outputs(got), bsw_gen.outputs(want))


def test_hip_bsw_scores_beyond_14_bits_and_mixed_lengths(ctx):
    a, ra, qa = bsw_gen.make_pairs(300, seed=21, max_q=120, h0_max=30000)      # h0 + qlen >= 2^14 for most pairs
    b, rb, qb = bsw_gen.make_pairs(300, seed=22, max_q=700)                     # both kernels in one batch
    b = b.copy(); b["idr"] += ra.shape[0]; b["idq"] += qa.shape[0]
    pairs = np.concatenate([a, b]); ref = np.concatenate([ra, rb]); qer = np.concatenate([qa, qb])
    order = np.random.default_rng(3).permutation(pairs.shape[0])
    pairs = pairs[order].copy()
    want = pairs.copy()
    O.bsw_batch(want, ref, qer, 100, O.default_bsw_params(5), threads=0)
    got = pairs.copy()
    ctx.bsw_batch(got, ref, qer, 100, _opt(5))
    assert np.array_equal(bsw_gen.outputs(got), bsw_gen.outputs(want))


def test_hip_bsw_queries_beyond_the_lds_resident_limit(ctx):
    """getScores16's class reaches 32 k bases (src/bandedSWA.h:47-86): queries of 5-9 k bases keep their DP rows in an HBM
    workspace instead of LDS; a mixed batch (short pairs alongside) exercises both storage modes of one launch sequence."""
    a, ra, qa = bsw_gen.make_pairs(6, seed=31, min_q=5000, max_q=9000, h0_max=100)
    b, rb, qb = bsw_gen.make_pairs(200, seed=32, max_q=300)
    b = b.copy(); b["idr"] += ra.shape[0]; b["idq"] += qa.shape[0]
    pairs = np.concatenate([a, b]); ref = np.concatenate([ra, rb]); qer = np.concatenate([qa, qb])
    for w in (100, 500):
        want = pairs.copy()
        O.bsw_batch(want, ref, qer, w, O.default_bsw_params(5), threads=0)
        got = pairs.copy()
        ctx.bsw_batch(got, ref, qer, w, _opt(5))
        assert np.array_equal(bsw_gen.outputs(got), bsw_gen.outputs(want)), w


@pytest.mark.parametrize("w", [100, 40, 200, 7])
def test_ring_of_band_columns_equals_a_word_per_query_column(w):
    """Round 6: queries longer than the band is wide keep 2w + 2 columns in a ring (k_bsw_lane_circ, tuning "bsw_circ", the default) instead of one LDS word
    per query column: same six outputs as the oracle and as the linear kernel, for every class the ring replaces -- long targets (the band slides the
    whole query length), large h0 (first-row values beyond the ring's first fill), gaps that re-grow the band, unrelated pairs (early exits)."""
    c = hipapi.Context(0)
    try:
        c.set_tuning("bsw_lane_min_pairs", 0)
        for kw in (dict(max_q=600, min_q=150), dict(max_q=600, min_q=230, sub=0.06, indel=0.03), dict(max_q=420, min_q=200, h0_max=900),
                   dict(max_q=320, min_q=100, sub=0.3, unrelated_frac=0.5), dict(max_q=600, min_q=1, h0_max=60)):
            pairs, ref, qer = bsw_gen.make_pairs(3000, seed=61 + w, **kw)
            want = pairs.copy()
            O.bsw_batch(want, ref, qer, w, O.default_bsw_params(5), threads=0)
            for circ in (1, 0):
                c.set_tuning("bsw_circ", circ)
                got = pairs.copy()
                c.bsw_batch(got, ref, qer, w, _opt(5))
                assert np.array_equal(bsw_gen.outputs(got), bsw_gen.outputs(want)), (kw, w, circ)
    finally:
        c.close()
