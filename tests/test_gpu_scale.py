"""Parity at the size the headline number is quoted on: an index of more than 2^32 suffixes (64-bit slot arithmetic in
the search kernel, grid-stride staging kernels, the 5-byte position image above 4 G entries), full seed dump against
the pinned oracle.  The index is built the way bench.py builds it -- suffix array, entries and P-RMI on the GPU -- and the
suffix array is verified here on the host (a permutation whose sampled neighbours are in suffix order) before the oracle uses it."""
import os

import numpy as np
import pytest

import oracle_py as O
from pymeme import hipapi, synth, workload

pytestmark = pytest.mark.gpu


def _enough_memory():
    try:
        import psutil
        return psutil.virtual_memory().available > 300e9
    except ImportError:
        return True


@pytest.mark.skipif(os.environ.get("MEME_SKIP_SCALE_TEST") == "1" or not _enough_memory(),
                    reason="needs ~300 GB of host memory for the suffix-array builder")
def test_seeds_equal_oracle_above_2_pow_32_suffixes():
    import torch
    l_pac = 2_150_000_000
    n = 2 * l_pac
    assert n > 1 << 32
    g = synth.make_genome(l_pac, seed=11)
    text = hipapi.fwd_rc_text(g)
    dev = torch.device("cuda", 0)
    ctx = hipapi.Context(0)
    try:
        d_text, d_sa = hipapi.build_sa_device(ctx, text)
        # a permutation of 0..n-1 ...
        seen = torch.zeros(n, dtype=torch.bool, device=dev)
        seen[d_sa] = True
        assert bool(seen.all())
        del seen
        sa = d_sa.cpu().numpy().view(np.uint64)
        # ... whose neighbours are in suffix order (a suffix that ends sorts as if followed by T's), sampled at the ends, around
        # slot 2^32 and at random
        rng = np.random.default_rng(5)
        slots = np.concatenate([np.arange(0, 400), np.arange((1 << 32) - 200, (1 << 32) + 200), np.arange(n - 401, n - 1),
                                rng.integers(0, n - 1, size=3000)])

        def suffix(p, m):
            s_ = text[p:p + m]
            return s_ if s_.shape[0] == m else np.concatenate([s_, np.full(m - s_.shape[0], 3, np.uint8)])

        for i in slots:
            a, b = int(sa[i]), int(sa[i + 1])
            m = 64
            while True:
                x, y = suffix(a, m), suffix(b, m)
                if not np.array_equal(x, y) or m > 1 << 16:
                    break
                m *= 4
            d = np.nonzero(x != y)[0]
            if d.size == 0:
                assert a > b, (int(i), a, b)          # both run into the end of the text: the shorter suffix first
            else:
                assert x[d[0]] < y[d[0]], (int(i), a, b)
        d_pos5 = hipapi.pos5_from_sa_torch(ctx, d_sa, n)
        # the 5-byte image equals the reference's on-disk encoding, also above 2^32 entries
        for lo in (0, (1 << 32) - 500, n - 1000):
            chunk = d_pos5[lo * 5:(lo + 1000) * 5].cpu().numpy().reshape(-1, 5)
            pos = (chunk[:, :4].copy().view("<u4")[:, 0].astype(np.uint64) << np.uint64(8)) | chunk[:, 4].astype(np.uint64)
            assert np.array_equal(pos, sa[lo:lo + 1000]), lo
        del d_sa
        torch.cuda.empty_cache()
        d_pac, d_ent0 = hipapi.stage_entries_torch(ctx, n, d_text, d_pos5)
        d_l2, n_l2, d_l1, n_l1 = hipapi.train_prmi_device(ctx, d_ent0, n, 24)
        keep = (d_pac, d_ent0) + hipapi.attach_index_torch(ctx, n, d_pac, d_ent0, d_l2, n_l2, d_l1, n_l1)
        d_ent = keep[1].view(-1, 2)
        # entries: keys sorted, positions = the suffix array, and each key the first 32 bases of the suffix its slot points to
        # (sampled around 2^32 and at the ends)
        for lo in (0, (1 << 32) - 500, n - 1000):
            e = d_ent[lo:lo + 1000].cpu().numpy()
            k = e[:, 0].view(np.uint64)
            assert np.all(k[1:] >= k[:-1]), lo
            assert np.array_equal(e[:, 1].view(np.uint64), sa[lo:lo + 1000]), lo
            for i in (0, 499, 999):
                p = int(sa[lo + i])
                s32 = text[p:p + 32] if p + 32 <= n else np.concatenate([text[p:], np.full(32 - (n - p), 3, np.uint8)])
                want = 0
                for b in s32:
                    want = (want << 2) | int(b & 3)
                assert int(k[i]) == want, (lo, i)
        idx = O.Index(text, sa)
        for L, nreads, kw in ((150, 3000, {}), (250, 600, dict(sub_rate=0.05))):
            reads = workload.make_reads_fast(g, nreads, L, seed=12 + L, **kw)
            off = np.arange(0, (nreads + 1) * L, L, dtype=np.int64)
            smems, so, hits, ho = ctx.seed_batch(reads, off)
            slots, counts, hl = hipapi.smems_to_slots(smems, so, hits, ho)
            sm, ns, oh, nh, _ = O.seed_batch(idx, reads, off, smem_cap=4096, hit_cap=1 << 17, threads=0)
            assert O.format_seed_dump(slots, counts, hl) == O.format_seed_dump(sm, ns, oh), L
    finally:
        ctx.close()
