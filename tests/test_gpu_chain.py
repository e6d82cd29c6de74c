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


def _run(ctx, prefix, reads, wave_tiers=1):
    ctx.set_tuning("chain_wave_tiers", wave_tiers)
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


@pytest.mark.parametrize("wave_tiers", [1, 0])
def test_device_chains_equal_reference_golden_and_oracle(tmp_path, wave_tiers):
    """wave_tiers 1: repeat-rich reads go through the LDS tier (one wavefront per read, chains in LDS), the B-tree tier takes what that
    leaves; 0: everything beyond the lane-per-read tier goes through the B-tree tier."""
    g, reads = chain_golden_workload()
    fa = str(tmp_path / "c.fa")
    synth.write_fasta(fa, g, name="cg", contigs=3)
    prefix = build_index(fa, bits=14)
    G = np.load(os.path.join(GOLDEN, "chain_golden.npz"))
    ctx = hipapi.Context(0)
    try:
        (smems, smem_off, hits, hit_off), R, l_pac, contigs = _run(ctx, prefix, reads, wave_tiers)
    finally:
        ctx.close()
    n = len(reads)
    assert n == G["read_len"].shape[0] and l_pac == int(G["l_pac"])
    # the seeds are the fixture's inputs (the fixture was dumped from the same backend; the seeds themselves are pinned by test_gpu_seed).
    # The ORDER of a read's SMEMs is not part of the contract -- the consumer sorts them by (start, end), src/bwamem.cpp:1397, and equal
    # keys carry equal hit lists -- so the comparison is per read on the sorted (start, end, hits) triples.
    assert np.array_equal(smem_off, G["smem_off"]) and np.array_equal(hit_off, G["hit_off"])

    def canon(sm_start, sm_end, sm_hb, sm_hc, hv):
        return sorted((int(a), int(b), tuple(int(x) for x in hv[int(hb):int(hb) + int(hc)])) for a, b, hb, hc in zip(sm_start, sm_end, sm_hb, sm_hc))
    for r in range(n):
        s0, s1, h0, h1 = int(smem_off[r]), int(smem_off[r + 1]), int(hit_off[r]), int(hit_off[r + 1])
        mine = canon(smems["start"][s0:s1], smems["end"][s0:s1], smems["hitbeg"][s0:s1], smems["hitcount"][s0:s1], hits[h0:h1])
        gs = G["smems"][s0:s1]
        assert mine == canon(gs[:, 0], gs[:, 1], gs[:, 2], gs[:, 3], G["hits"][h0:h1]), r
    opt = O.default_chain_opt(l_pac)
    contig_off = np.array([c[0] for c in contigs], np.int64)
    contig_alt = np.zeros(len(contigs), np.uint8)
    assert R["n_fallback"] == 0 and not R["fallback"].any()                 # every read is chained on the device
    for r in range(n):
        c0, c1 = int(G["chain_off"][r]), int(G["chain_off"][r + 1])
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
    # and the oracle agrees with the device on the same seeds, read by read, field by field
    read_len = np.array([len(r) for r in reads], np.int32)
    assert O.chain_compare_batch(smems, smem_off, hits, hit_off, read_len, contig_off, contig_alt, opt, R) == (0, -1)
    assert R["n_tier2"] > 0                                                   # the fixture's repeat reads went through the wavefront-per-read tier


def _contigs3():
    return [(0, 70_000, 0), (70_000, 80_000, 0), (150_000, 50_000, 1)]


def test_device_chains_equal_reference_on_equal_positions_and_big_trees():
    """tests/golden/chain_dup_golden.npz through meme_chain_batch_host: made-up seed sets with up to 1 300 chains per read, most with
    several chains at EQUAL positions, against the chains the compiled reference (its B-tree, its introsort) made of them."""
    import chain_gen  # noqa: F401  (the fixture's generator; here only the fixture is used)
    G = np.load(os.path.join(GOLDEN, "chain_dup_golden.npz"))
    sm = np.zeros(G["smems"].shape[0], hipapi.MEM_TL)
    sm["start"], sm["end"], sm["hitbeg"], sm["hitcount"] = G["smems"].T
    ctx = hipapi.Context(0)
    try:
        R = ctx.chain_batch_host(sm, G["smem_off"], G["hits"], G["hit_off"], G["read_len"], _contigs3(), hipapi.default_chain_opt(int(G["l_pac"])))
    finally:
        ctx.close()
    assert R["n_fallback"] == 0
    assert np.array_equal(R["chain_off"], G["chain_off"]) and np.array_equal(R["seed_off"], G["seed_off"])
    assert np.array_equal(R["tree_size"], G["tree_size"])
    for k, f in enumerate(("pos", "rid", "n_seeds", "w", "kept", "first", "is_alt", "seed_beg")):
        assert np.array_equal(R["chains"][f].astype(np.int64), G["chains"][:, k]), f
    assert np.array_equal(np.stack([R["seeds"]["rbeg"], R["seeds"]["qbeg"], R["seeds"]["len"]], 1), G["seeds"])
    has = np.diff(G["chain_off"]) > 0
    assert np.array_equal(R["frac_rep"].view(np.uint32)[has], G["frac_rep_bits"][has])


def _adversarial(n, seed):
    import chain_gen
    reads = chain_gen.workload(seed, n, l_pac=200_000)
    smem_off = np.zeros(len(reads) + 1, np.int64); hit_off = np.zeros(len(reads) + 1, np.int64)
    for r, (sm, h) in enumerate(reads):
        smem_off[r + 1] = smem_off[r] + sm.shape[0]; hit_off[r + 1] = hit_off[r] + h.shape[0]
    smems = np.concatenate([sm for sm, _ in reads]).astype(hipapi.MEM_TL)
    hits = np.concatenate([h for _, h in reads])
    read_len = np.array([250 if r % 5 == 2 else 150 for r in range(len(reads))], np.int32)
    return smems, smem_off, hits, hit_off, read_len


def test_device_chains_equal_oracle_on_adversarial_batch():
    """5 000 made-up reads (tests/chain_gen.py) chained on the device and compared with the oracle by the batched checker."""
    smems, smem_off, hits, hit_off, read_len = _adversarial(5000, 991)
    ctx = hipapi.Context(0)
    try:
        R = ctx.chain_batch_host(smems, smem_off, hits, hit_off, read_len, _contigs3(), hipapi.default_chain_opt(200_000))
    finally:
        ctx.close()
    assert R["n_fallback"] == 0 and R["n_tier2"] > 1000
    bad = O.chain_compare_batch(smems, smem_off, hits, hit_off, read_len, np.array([0, 70_000, 150_000], np.int64), np.array([0, 0, 1], np.uint8),
                                O.default_chain_opt(200_000), R)
    assert bad == (0, -1), bad


@pytest.mark.parametrize("name,change", [("narrow_band", dict(w=3)), ("wide_band", dict(w=2000)), ("short_gap", dict(max_chain_gap=40)),
                                         ("few_hits", dict(max_occ=37)), ("strict_filter", dict(drop_ratio=0.9, mask_level=0.1)),
                                         ("few_extended", dict(max_chain_extend=3)), ("weight_floor", dict(min_chain_weight=40))])
def test_device_chains_equal_oracle_under_other_options(name, change):
    """The same kind of batch under other chaining options: the band and gap limits enter test_and_merge and with it the rule by which the
    wavefront tier commits a batch of hits at once; max_occ the sampling of the hits; the rest the filter."""
    smems, smem_off, hits, hit_off, read_len = _adversarial(2500, 1200 + len(name))
    do, oo = hipapi.default_chain_opt(200_000), O.default_chain_opt(200_000)
    for k, v in change.items():
        setattr(do, k, v); setattr(oo, k, v)
    ctx = hipapi.Context(0)
    try:
        R = ctx.chain_batch_host(smems, smem_off, hits, hit_off, read_len, _contigs3(), do)
    finally:
        ctx.close()
    assert R["n_fallback"] == 0
    bad = O.chain_compare_batch(smems, smem_off, hits, hit_off, read_len, np.array([0, 70_000, 150_000], np.int64), np.array([0, 0, 1], np.uint8), oo, R)
    assert bad == (0, -1), (name, bad)


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
        with pytest.raises(hipapi.MemeError, match="not a valid reference sequence"):          # a bntann1_t length is positive and inside the genome
            ctx.chain_last_batch_host([(0, 70_000, 0)], hipapi.default_chain_opt(60_000))
        res = ctx.chain_last_batch_host([(0, 60_000, 0)], hipapi.default_chain_opt(60_000))
        assert res["chain_off"].shape[0] == 51 and res["chain_off"][-1] == res["chains"].shape[0] >= 40
        w = res["chains"]["w"]
        for i in range(50):                                     # the filter's output is ordered by weight
            c = w[int(res["chain_off"][i]):int(res["chain_off"][i + 1])]
            assert np.all(c[:-1] >= c[1:])
    finally:
        ctx.close()
