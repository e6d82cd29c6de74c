"""P-RMI training on the device (meme_prmi_train_device) against the host trainer: the two parameter tables must be equal
bit for bit (same double arithmetic), for models without and with a partial third layer."""
import numpy as np
import pytest
import torch

from pymeme import hipapi, hostapi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hipapi.Context(0)
    yield c
    c.close()


def _compare(ctx, fwd, bits, threshold=1000):
    text, sa = hostapi.build_sa(fwd)
    n = int(text.shape[0])
    l1, l2 = hostapi.train_prmi(text, sa, bits=bits, partial_threshold=threshold)
    dev = torch.device("cuda", 0)
    d_text = torch.from_numpy(text).to(dev)
    d_sa = torch.from_numpy(sa.view(np.int64)).to(dev)
    d_pos5 = hipapi.pos5_from_sa_torch(ctx, d_sa, n)
    d_pac, d_ent = hipapi.stage_entries_torch(ctx, n, d_text, d_pos5)
    d_l2, n_l2, d_l1, n_l1 = hipapi.train_prmi_device(ctx, d_ent, n, bits, threshold)
    g2 = d_l2.cpu().numpy().view(hostapi.RMI_DTYPE)
    g1 = d_l1.cpu().numpy().view(hostapi.RMI_DTYPE)[:n_l1]
    if n_l1 == 0 and l1.shape[0] == 1:
        assert not l1.view(np.uint64).any()                   # the host trainer pads an empty third layer with one zero record
        l1 = l1[:0]
    assert n_l2 == l2.shape[0] and n_l1 == l1.shape[0], (n_l2, l2.shape, n_l1, l1.shape)
    for name, got, want in (("second layer", g2, l2), ("third layer", g1, l1)):
        same = got.view(np.uint64).reshape(-1, 3) == want.view(np.uint64).reshape(-1, 3)
        if not same.all():
            bad = np.nonzero(~same.all(axis=1))[0]
            raise AssertionError("%s: %d of %d records differ, first %d: %r vs %r" % (name, bad.size, want.shape[0], bad[0], got[bad[0]], want[bad[0]]))
    return n_l1


def test_plain_leaves(ctx):
    assert _compare(ctx, synth.make_genome(300_000, seed=5, repeat_frac=0.05), bits=16) == 0


def test_partial_third_layer_everywhere(ctx):
    # 2^8 leaves over 800 k keys: every leaf is beyond the threshold and routes into third-layer records
    assert _compare(ctx, synth.make_genome(400_000, seed=6, repeat_frac=0.1, n_dups=6, dup_len=2000), bits=8) > 30000


def test_mixed_and_equal_key_runs(ctx):
    # homopolymers and exact duplicates: long runs of equal 32-base keys, leaves whose keys are all identical, empty leaves
    fwd = synth.make_genome(500_000, seed=7, repeat_frac=0.2, repeat_len=300, n_families=3, divergence=0.0, n_dups=12, dup_len=4000, poly_runs=10)
    fwd[1000:9000] = 0                                          # 8 000 A in a row (and as many T on the other strand)
    assert _compare(ctx, fwd, bits=14, threshold=200) > 0


def test_low_threshold_small_text(ctx):
    rng = np.random.default_rng(8)
    assert _compare(ctx, rng.integers(0, 4, size=5000).astype(np.uint8), bits=4, threshold=50) > 0
