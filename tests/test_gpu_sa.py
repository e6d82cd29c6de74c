"""Device-side suffix-array construction (meme_sa_build_device) against the host builder, whose output is verified
byte-identical to `bwa-meme index -a meme` (tests/test_ref_live.py, where the compiled reference is available)."""
import numpy as np
import pytest

from pymeme import hipapi, hostapi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hipapi.Context(0)
    yield c
    c.close()


def _check(ctx, fwd):
    text, sa = hostapi.build_sa(fwd)
    t2 = hipapi.fwd_rc_text(fwd)
    assert np.array_equal(text, t2)
    _, d_sa = hipapi.build_sa_device(ctx, t2)
    got = d_sa.cpu().numpy().view(np.uint64)
    if not np.array_equal(got, sa):
        bad = np.nonzero(got != sa)[0]
        raise AssertionError("suffix arrays differ at %d of %d slots, first at %d: %d vs %d" % (bad.size, sa.size, bad[0], got[bad[0]], sa[bad[0]]))


def test_random_genome(ctx):
    _check(ctx, synth.make_genome(200_000, seed=3, repeat_frac=0.0, n_dups=0, poly_runs=0))


def test_repeat_rich_genome(ctx):
    # long exact duplicates and homopolymer runs: many rounds of prefix doubling, ties that end at the end of the text
    _check(ctx, synth.make_genome(600_000, seed=31, repeat_frac=0.2, repeat_len=300, n_families=4, divergence=0.02, n_dups=10,
                                  dup_len=3000, poly_runs=8))


@pytest.mark.parametrize("tail", ["A", "T", "ACGT"])
def test_text_ends_in_a_run(ctx, tail):
    # the suffixes of a homopolymer at the very end of the text differ only in their length: shorter sorts first
    rng = np.random.default_rng(5)
    body = rng.integers(0, 4, size=4000).astype(np.uint8)
    codes = {"A": 0, "C": 1, "G": 2, "T": 3}
    t = np.array([codes[c] for c in tail] * (120 // len(tail)), np.uint8)
    # fwd + rc ends with the reverse complement of the START of the forward strand: put the run there too
    rc_t = (3 - t[::-1]).astype(np.uint8)
    _check(ctx, np.concatenate([rc_t, body, t]))


def test_smallest_text(ctx):
    _check(ctx, np.array([0, 1, 2, 3] * 8, np.uint8))           # 64 suffixes


def test_midsize_genome_equals_host_builder(ctx):
    _check(ctx, synth.make_genome(16_000_000, seed=91, repeat_frac=0.05, repeat_len=400, n_families=8, divergence=0.03,
                                  n_dups=20, dup_len=5000, poly_runs=6))
