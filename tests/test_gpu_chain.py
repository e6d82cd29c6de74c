"""Chaining on the device (meme_chain_last_batch_host = mem_chain_Learned + mem_chain_flt for the batch just seeded), through
the C ABI, against the chains the compiled reference made of the same reads (tests/golden/chain_golden.npz) and against the
oracle's restatement on every read."""
import os

import numpy as np
import pytest

import oracle_py as O
from common import GOLDEN, build_index, chain_golden_workload
from pymeme import hipapi, synth

pytestmark = pytest.mark.gpu


def _run(ctx, prefix, reads):
    off = np.zeros(len(reads) + 1, np.int64)
    off[1:] = np.cumsum([len(r) for r in reads])
    flat = np.concatenate(reads)
    ctx.load_index_files(prefix)
    smems, smem_off, hits, hit_off = ctx.seed_batch_host(flat, off)
    ann = [l.split() for l in open(prefix + ".ann")]
    l_pac = int(ann[0][0])
    contigs = [(int(ann[2 + 2 * k][0]), int(ann[2 + 2 * k][1]), 0) for k in range(int(ann[0][1]))]
    res = ctx.chain_last_batch_host(contigs, hipapi.default_chain_opt(l_pac))
    return (smems, smem_off, hits, hit_off), res, l_pac, contigs


def test_device_chains_equal_reference_golden_and_oracle(tmp_path):
    g, reads = chain_golden_workload()
    fa = str(tmp_path / "c.fa")
    synth.write_fasta(fa, g, name="cg", contigs=3)
    prefix = build_index(fa, bits=14)
    G = np.load(os.path.join(GOLDEN, "chain_golden.npz"))
    ctx = hipapi.Context(0)
    try:
        (smems, smem_off, hits, hit_off), R, l_pac, contigs = _run(ctx, prefix, reads)
    finally:
        ctx.close()
    n = len(reads)
    assert n == G["read_len"].shape[0] and l_pac == int(G["l_pac"])
    # the seeds are the fixture's inputs (the fixture was dumped from the same backend; the seeds themselves are pinned by test_gpu_seed)
    assert np.array_equal(smem_off, G["smem_off"]) and np.array_equal(hits, G["hits"])
    opt = O.default_chain_opt(l_pac)
    contig_off = np.array([c[0] for c in contigs], np.int64)
    contig_alt = np.zeros(len(contigs), np.uint8)
    n_dev = 0
    for r in range(n):
        c0, c1 = int(G["chain_off"][r]), int(G["chain_off"][r + 1])
        if R["fallback"][r]:
            assert R["chain_off"][r + 1] == R["chain_off"][r]
            continue
        n_dev += 1
        d0, d1 = int(R["chain_off"][r]), int(R["chain_off"][r + 1])
        assert d1 - d0 == c1 - c0 and int(R["tree_size"][r]) == int(G["tree_size"][r]), r
        if d1 > d0:
            assert R["frac_rep"][r:r + 1].view(np.uint32)[0] == G["frac_rep_bits"][r], r
        sd = R["seeds"][int(R["seed_off"][r]):int(R["seed_off"][r + 1])]
        for k in range(d1 - d0):
            ch = R["chains"][d0 + k]
            want = G["chains"][c0 + k]
            got = [int(ch[f]) for f in ("pos", "rid", "n_seeds", "w", "kept", "first", "is_alt")]
            assert got == [int(x) for x in want[:7]], (r, k, got, want)
            gs = sd[int(ch["seed_beg"]):int(ch["seed_beg"]) + int(ch["n_seeds"])]
            ws = G["seeds"][int(want[7]):int(want[7]) + int(want[2])]
            assert np.array_equal(np.stack([gs["rbeg"], gs["qbeg"], gs["len"]], 1), ws), (r, k)
        # and the oracle agrees with the device on the same seeds
        sm = smems[int(smem_off[r]):int(smem_off[r + 1])]
        rc, och, osd, tree, frac = O.chain_read(sm, hits[int(hit_off[r]):int(hit_off[r + 1])], len(reads[r]), contig_off, contig_alt, opt)
        assert rc == d1 - d0 and tree == int(R["tree_size"][r]), r
    assert n_dev > 0.95 * n and R["n_fallback"] == n - n_dev, (n_dev, n)


def test_chain_call_needs_a_seeded_batch_and_sane_options(tmp_path):
    g = synth.make_genome(60_000, seed=9)
    fa = str(tmp_path / "e.fa")
    synth.write_fasta(fa, g, contigs=1)
    prefix = build_index(fa, bits=12)
    ctx = hipapi.Context(0)
    try:
        ctx.load_index_files(prefix)
        with pytest.raises(hipapi.MemeError, match="no seeded batch"):
            ctx.chain_last_batch_host([(0, 60_000, 0)], hipapi.default_chain_opt(60_000))
        r, _, _ = synth.make_reads(g, 50, 100, seed=10)
        ctx.seed_batch_host(r.reshape(-1), np.arange(0, 51 * 100, 100, dtype=np.int64))
        bad = hipapi.default_chain_opt(60_000)
        bad.max_occ = 0
        with pytest.raises(hipapi.MemeError, match="bad options"):
            ctx.chain_last_batch_host([(0, 60_000, 0)], bad)
        res = ctx.chain_last_batch_host([(0, 60_000, 0)], hipapi.default_chain_opt(60_000))
        assert res["chain_off"].shape[0] == 51 and res["chain_off"][-1] == res["chains"].shape[0] >= 40
        w = res["chains"]["w"]
        for i in range(50):                                     # the filter's output is ordered by weight
            c = w[int(res["chain_off"][i]):int(res["chain_off"][i + 1])]
            assert np.all(c[:-1] >= c[1:])
    finally:
        ctx.close()
