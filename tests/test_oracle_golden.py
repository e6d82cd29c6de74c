"""Pins the C oracle (and our index builder) against golden vectors produced by the COMPILED
REFERENCE (tests/golden/make_golden.py): seed dumps of learned_seeding_big_read and the six outputs of
scalarBandedSWAWrapper.  CPU only."""
import os

import numpy as np
import pytest

import bsw_gen
import oracle_py as O
from common import GOLDEN, build_index, read_fastq_codes


@pytest.fixture(scope="module")
def g1_index():
    return O.load_index_files(build_index(os.path.join(GOLDEN, "g1.fa")))


@pytest.mark.parametrize("length", [150, 250, 60, 25])
def test_oracle_seeds_equal_reference_dump(g1_index, length):
    reads, off = read_fastq_codes(os.path.join(GOLDEN, "g1_reads_%d.fq" % length))
    sm, ns, hits, nh, _ = O.seed_batch(g1_index, reads, off, smem_cap=256, hit_cap=4096, threads=2)
    dump = O.format_seed_dump(sm, ns, hits)
    want = open(os.path.join(GOLDEN, "g1_seeds_%d.txt" % length)).read()
    assert dump == want


@pytest.mark.parametrize("w,eb", [(100, 5), (100, 0), (200, 5), (200, 0)])
def test_oracle_bsw_equals_reference_scalar(w, eb):
    z = np.load(os.path.join(GOLDEN, "bsw_golden.npz"))
    pairs = z["pairs"].copy()
    O.bsw_batch(pairs, z["ref"], z["qer"], w, O.default_bsw_params(end_bonus=eb), threads=2)
    assert np.array_equal(bsw_gen.outputs(pairs), z["scalar_w%d_eb%d" % (w, eb)])


def test_reference_simd16_agrees_with_scalar_where_it_matters():
    """The reference's AVX-512 kernel deviates from its scalar twin only in (gtle, gscore) of pairs whose
    end-to-end score cannot win; the values mem_chain2aln commits (src/bwamem.cpp:3020-3036: local vs
    to-end decision, query/ref ends, truesc) are identical."""
    z = np.load(os.path.join(GOLDEN, "bsw_golden.npz"))
    a, b = z["scalar_w100_eb5"], z["simd16_w100_eb5"]
    pen_clip = 5

    def committed(o):
        score, tle, gtle, qle, gscore = o[:, 0], o[:, 1], o[:, 2], o[:, 3], o[:, 4]
        local = (gscore <= 0) | (gscore <= score - pen_clip)
        return np.where(local, qle, -1), np.where(local, tle, gtle), np.where(local, score, gscore)

    for x, y in zip(committed(a), committed(b)):
        assert np.array_equal(x, y)
    assert np.array_equal(a[:, 0], b[:, 0]) and np.array_equal(a[:, 5], b[:, 5])


@pytest.mark.parametrize("length", [150, 250, 60, 25])
def test_refpath_restatement_agrees_with_semantic_oracle(g1_index, length):
    """The instrumented probe-sequence restatement (work counters for the roofline) finds the same SMEMs."""
    prefix = build_index(os.path.join(GOLDEN, "g1.fa"))
    l1, l2 = O.load_prmi_files(prefix)
    reads, off = read_fastq_codes(os.path.join(GOLDEN, "g1_reads_%d.fq" % length))
    sm, ns, hits, nh, _ = O.seed_batch(g1_index, reads, off, smem_cap=256, hit_cap=4096, threads=2)
    rsm, rns, ctr = O.refpath_seed_batch(g1_index, l1, l2, reads, off, smem_cap=256, threads=2)
    assert np.array_equal(ns, rns)
    for r in range(ns.shape[0]):
        k = ns[r]
        for f in ("start", "end", "hitcount"):
            assert np.array_equal(sm[r, :k][f], rsm[r, :k][f]), (r, f)
    assert ctr["compares"] > ctr["lookups"] > 0 and ctr["smems"] == int(ns.sum())


def test_chain_oracle_equals_reference_golden():
    """orc_chain_read (restatement of mem_chain_Learned + test_and_merge + mem_chain_weight + mem_chain_flt, klib sort order
    included) against the chains the compiled reference made of the same seeds (tests/golden/chain_golden.npz, 2 800 reads of a
    repeat-rich genome, 689 of them with several chains; generator: tests/golden/make_chain_golden.py)."""
    import numpy as np
    G = np.load(os.path.join(GOLDEN, "chain_golden.npz"))
    opt = O.default_chain_opt(int(G["l_pac"]))
    contig_alt = np.zeros(G["contig_off"].shape[0], np.uint8)
    for r in range(G["read_len"].shape[0]):
        s0, s1 = int(G["smem_off"][r]), int(G["smem_off"][r + 1])
        sm = np.zeros(s1 - s0, O.MEM_TL_DTYPE)
        sm["start"], sm["end"], sm["hitbeg"], sm["hitcount"] = G["smems"][s0:s1].T
        hits = G["hits"][int(G["hit_off"][r]):int(G["hit_off"][r + 1])]
        rc, ch, sd, tree, frac = O.chain_read(sm, hits, int(G["read_len"][r]), G["contig_off"], contig_alt, opt)
        c0, c1 = int(G["chain_off"][r]), int(G["chain_off"][r + 1])
        want = G["chains"][c0:c1]
        assert rc == c1 - c0 and tree == int(G["tree_size"][r]), (r, rc, c1 - c0, tree, int(G["tree_size"][r]))
        if rc:
            assert np.float32(frac).view(np.uint32) == G["frac_rep_bits"][r], r
        for k in range(rc):
            got = [int(ch[k][f]) for f in ("pos", "rid", "n_seeds", "w", "kept", "first", "is_alt")]
            assert got == [int(x) for x in want[k][:7]], (r, k, got, want[k])
            b = int(want[k][7])
            ws = G["seeds"][b:b + int(want[k][2])]
            gs = sd[int(ch[k]["seed_beg"]):int(ch[k]["seed_beg"]) + int(ch[k]["n_seeds"])]
            assert np.array_equal(np.stack([gs["rbeg"], gs["qbeg"], gs["len"]], 1), ws), (r, k)


def test_chain_oracle_equals_reference_golden_equal_positions():
    """The same on tests/golden/chain_dup_golden.npz: 300 made-up reads (tests/chain_gen.py) with up to 1 300 chains, most with several
    chains at EQUAL positions -- their order is decided by the reference's B-tree (src/kbtree.h: where kb_putp places an equal key, which
    one kb_intervalp finds), restated in the oracle; expected chains = the compiled reference's (tests/golden/make_chain_dup_golden.py)."""
    import numpy as np
    G = np.load(os.path.join(GOLDEN, "chain_dup_golden.npz"))
    opt = O.default_chain_opt(int(G["l_pac"]))
    n_dup = 0
    for r in range(G["read_len"].shape[0]):
        s0, s1 = int(G["smem_off"][r]), int(G["smem_off"][r + 1])
        sm = np.zeros(s1 - s0, O.MEM_TL_DTYPE)
        sm["start"], sm["end"], sm["hitbeg"], sm["hitcount"] = G["smems"][s0:s1].T
        hits = G["hits"][int(G["hit_off"][r]):int(G["hit_off"][r + 1])]
        rc, ch, sd, tree, frac = O.chain_read(sm, hits, int(G["read_len"][r]), G["contig_off"], G["contig_alt"], opt, chain_cap=8192, seed_cap=1 << 17)
        c0, c1 = int(G["chain_off"][r]), int(G["chain_off"][r + 1])
        want = G["chains"][c0:c1]
        assert rc == c1 - c0 and tree == int(G["tree_size"][r]), (r, rc, c1 - c0, tree)
        if rc:
            assert np.float32(frac).view(np.uint32) == G["frac_rep_bits"][r], r
            got = np.stack([ch[k] for k in ("pos", "rid", "n_seeds", "w", "kept", "first", "is_alt", "seed_beg")], 1).astype(np.int64)
            assert np.array_equal(got, want), r
            assert np.array_equal(np.stack([sd["rbeg"], sd["qbeg"], sd["len"]], 1), G["seeds"][int(G["seed_off"][r]):int(G["seed_off"][r + 1])]), r
            n_dup += len(np.unique(want[:, 0])) < rc
    assert n_dup > 60, n_dup


def test_chain_oracle_edge_cases():
    """orc_chain_read on hand-made inputs: no seeds, a read shorter than min_seed_len, two chains forced onto one position, a seed that bridges two contigs (dropped), merging of collinear seeds, too small an output (-2)."""
    import numpy as np
    opt = O.default_chain_opt(10_000)
    off = np.array([0, 5_000], np.int64)
    alt = np.zeros(2, np.uint8)

    def smem(start, end, hitbeg, hitcount):
        a = np.zeros(1, O.MEM_TL_DTYPE)
        a["start"], a["end"], a["hitbeg"], a["hitcount"] = start, end, hitbeg, hitcount
        return a
    none = np.zeros(0, O.MEM_TL_DTYPE)
    rc, ch, sd, tree, frac = O.chain_read(none, np.zeros(0, np.uint64), 100, off, alt, opt)
    assert (rc, tree) == (0, 0)
    rc, ch, sd, tree, frac = O.chain_read(smem(0, 15, 0, 1), np.array([100], np.uint64), 15, off, alt, opt)     # len < min_seed_len
    assert (rc, tree) == (0, 0)
    # one seed: one chain of weight = seed length
    rc, ch, sd, tree, frac = O.chain_read(smem(10, 40, 0, 1), np.array([1000], np.uint64), 100, off, alt, opt)
    assert rc == 1 and tree == 1 and int(ch[0]["w"]) == 30 and int(ch[0]["pos"]) == 1000 and int(ch[0]["rid"]) == 0 and int(ch[0]["kept"]) == 3
    # two collinear seeds merge into one chain; a third on the other contig makes a second chain
    sm = np.concatenate([smem(0, 30, 0, 1), smem(40, 70, 1, 1), smem(72, 100, 2, 1)])
    rc, ch, sd, tree, frac = O.chain_read(sm, np.array([1000, 1040, 6000], np.uint64), 100, off, alt, opt)
    assert rc == 2 and tree == 2 and sorted(int(x) for x in ch["n_seeds"]) == [1, 2] and int(ch[0]["w"]) == 60     # heavier chain first
    # a seed across the contig boundary at 5 000 is dropped (bns_intv2rid < 0)
    rc, ch, sd, tree, frac = O.chain_read(smem(0, 30, 0, 1), np.array([4990], np.uint64), 100, off, alt, opt)
    assert (rc, tree) == (0, 0)
    # the same position twice with query offsets too far apart to merge: two chains on one B-tree key, the later one behind the first
    sm = np.concatenate([smem(0, 20, 0, 1), smem(150, 170, 1, 1)])
    rc, ch, sd, tree, frac = O.chain_read(sm, np.array([2000, 2000], np.uint64), 200, off, alt, opt)
    assert rc == 2 and tree == 2 and [int(x) for x in ch["pos"]] == [2000, 2000]
    # capacity
    sm = np.concatenate([smem(0, 30, 0, 1), smem(60, 90, 1, 1)])
    rc, ch, sd, tree, frac = O.chain_read(sm, np.array([1000, 3000], np.uint64), 100, off, alt, opt, chain_cap=1)
    assert rc == -2


def _fixture_as_device_result(G):
    """The reference's chains of a fixture in the layout the device returns (meme_chain / meme_chain_seed records)."""
    import numpy as np
    from pymeme import hipapi
    n = G["read_len"].shape[0]
    ch = np.zeros(G["chains"].shape[0], hipapi.CHAIN)
    for k, f in enumerate(("pos", "rid", "n_seeds", "w", "kept", "first", "is_alt")):
        ch[f] = G["chains"][:, k]
    seed_off = G["seed_off"] if "seed_off" in G else None
    if seed_off is None:                                     # chain_golden.npz stores a global seed_beg per chain
        seed_off = np.zeros(n + 1, np.int64)
        per_chain = G["chains"][:, 2]
        cs = np.concatenate([[0], np.cumsum(per_chain)])
        seed_off[1:] = cs[G["chain_off"][1:]]
        ch["seed_beg"] = G["chains"][:, 7] - np.repeat(seed_off[:-1], np.diff(G["chain_off"]))
    else:
        ch["seed_beg"] = G["chains"][:, 7]
    sd = np.zeros(G["seeds"].shape[0], hipapi.CHAIN_SEED)
    sd["rbeg"], sd["qbeg"], sd["len"] = G["seeds"].T
    return {"chains": ch, "seeds": sd, "chain_off": G["chain_off"], "seed_off": seed_off, "tree_size": G["tree_size"],
            "frac_rep": G["frac_rep_bits"].view(np.float32)}


def test_batched_chain_checker_accepts_the_reference_and_sees_a_flipped_field():
    """orc_chain_compare_batch (what bench.py and the -m gpu tests check whole batches with) on both fixtures: the reference's chains
    pass; one changed field, one changed seed and one changed tree size are each reported."""
    import numpy as np
    for name in ("chain_golden.npz", "chain_dup_golden.npz"):
        G = np.load(os.path.join(GOLDEN, name))
        res = _fixture_as_device_result(G)
        sm = np.zeros(G["smems"].shape[0], O.MEM_TL_DTYPE)
        sm["start"], sm["end"], sm["hitbeg"], sm["hitcount"] = G["smems"].T
        alt = G["contig_alt"] if "contig_alt" in G else np.zeros(G["contig_off"].shape[0], np.uint8)
        opt = O.default_chain_opt(int(G["l_pac"]))
        args = (sm, G["smem_off"], G["hits"], G["hit_off"], G["read_len"], G["contig_off"], alt, opt)
        assert O.chain_compare_batch(*args, res) == (0, -1), name
        r = int(np.nonzero(np.diff(G["chain_off"]) > 1)[0][3])
        bad = dict(res, chains=res["chains"].copy())
        bad["chains"]["w"][int(G["chain_off"][r]) + 1] += 1
        assert O.chain_compare_batch(*args, bad) == (1, r)
        bad = dict(res, seeds=res["seeds"].copy())
        bad["seeds"]["qbeg"][int(res["seed_off"][r])] += 1
        assert O.chain_compare_batch(*args, bad) == (1, r)
        bad = dict(res, tree_size=res["tree_size"].copy())
        bad["tree_size"][r] += 1
        assert O.chain_compare_batch(*args, bad) == (1, r)


def test_ext_oracle_equals_reference_golden():
    """orc_extend_batch (restatement of mem_chain2aln_across_reads_V2: job construction, banded SW, fold, band retry, purge) against the
    records the compiled reference made of the same reads and chains (tests/golden/ext_golden.npz; generator make_ext_golden.py):
    3 100 reads, 16 802 records, 10 208 of them purged, 443 extended with the doubled band."""
    import numpy as np
    from common import ext_golden_inputs
    I = ext_golden_inputs()
    G = np.load(os.path.join(GOLDEN, "ext_golden.npz"))
    assert np.array_equal(G["reg_off"], I["seed_off"])
    regs, (jobs, retried) = O.extend_batch(I["reads"], I["read_off"], I["chain_off"], I["chains"], I["seed_off"], I["seeds"], I["frac_rep"], I["text"],
                                           I["l_pac"], I["contig_off"], I["contig_len"])
    for k, f in enumerate(O.ALNREG_FIELDS):
        bad = np.nonzero(regs[f].astype(np.int64) != G["regs"][:, k])[0]
        assert bad.size == 0, (f, int(bad[0]), int(regs[f][bad[0]]), int(G["regs"][bad[0], k]))
    assert np.array_equal(regs["frac_rep"].view(np.uint32), G["frac_rep_bits"])
    assert retried > 300 and jobs > regs.shape[0]


def test_ksw_global2_oracle_equals_reference_golden():
    """orc_ksw_global2 against score and CIGAR of the compiled reference's ksw_global2 (tests/golden/gcig_golden.npz, 4 000 alignments:
    both strands, substitutions, indels up to 30 bases, N, bands 1..400, target shorter / longer than the query)."""
    import numpy as np
    from common import gcig_workload
    _, _, jobs, seqs = gcig_workload()
    G = np.load(os.path.join(GOLDEN, "gcig_golden.npz"))
    off = np.concatenate([[0], np.cumsum(G["n_cigar"])])
    for k, (J, (q, t)) in enumerate(zip(jobs, seqs)):
        sc, cg = O.ksw_global2(q, t, int(J["w"]))
        assert sc == int(G["score"][k]) and np.array_equal(cg, G["cigars"][off[k]:off[k + 1]]), (k, sc, int(G["score"][k]))


def test_gen_cigar2_oracle_equals_reference_golden():
    """orc_gen_cigar2 (bwa_gen_cigar2 whole) against score, CIGAR, NM and MD string of the compiled reference's function
    (tests/golden/gencig_golden.npz, 3 334 calls: both strands, the gap-free shortcut, band arguments 0..400, indels up to 30 bases, N)."""
    import numpy as np
    from common import gencig_workload
    from pymeme import hipapi
    g, reads, calls = gencig_workload()
    text = hipapi.fwd_rc_text(g)
    G = np.load(os.path.join(GOLDEN, "gencig_golden.npz"))
    off = np.concatenate([[0], np.cumsum(G["n_cigar"])])
    mds = G["md"].tobytes().split(b"\0")
    assert G["score"].shape[0] == calls.shape[0]
    for k, J in enumerate(calls):
        q = reads[int(J["read"])][int(J["qb"]):int(J["qb"]) + int(J["qlen"])]
        sc, cg, nm, md = O.gen_cigar2(text, g.shape[0], q, int(J["rb"]), int(J["rb"]) + int(J["tlen"]), int(J["w_"]))
        assert sc == int(G["score"][k]) and np.array_equal(cg, G["cigars"][off[k]:off[k + 1]]) and nm == int(G["nm"][k]) and md == mds[k], (k, sc, nm, md, mds[k])
    # calls the function rejects: empty spans, a target bridging the two strands
    l_pac = g.shape[0]
    assert O.gen_cigar2(text, l_pac, reads[0][:50], l_pac - 20, l_pac + 30, 10) is None
    assert O.gen_cigar2(text, l_pac, reads[0][:50], 100, 100, 10) is None


def test_aln2sam_oracle_equals_reference_golden():
    """orc_aln2sam against the text the compiled reference's mem_aln2sam wrote (tests/golden/sam_golden.npz: 1 500 records -- mapped / unmapped ends and
    mates, both strands, clipping styles, secondary flags, XA, with and without qualities -- hard and soft clipping, without and with a read group)."""
    import numpy as np
    from common import sam_workload
    recs, blob, names, reads, quals, contigs = sam_workload()
    cb, co = O.contig_table(contigs)
    G = np.load(os.path.join(GOLDEN, "sam_golden.npz"))
    for softclip, rg in ((0, b""), (1, b"grp1")):
        text, off = G["text_%d" % softclip].tobytes(), G["off_%d" % softclip]
        assert off.shape[0] == recs.shape[0] + 1
        for k in range(recs.shape[0]):
            got = O.aln2sam(recs[k], blob, names[k], reads[k], quals[k], cb, co, softclip, rg)
            assert got == text[off[k]:off[k + 1]], (k, softclip, got, text[off[k]:off[k + 1]])


def test_kswv_oracle_equals_reference_golden():
    """orc_kswv_batch against the kswr_t records of the compiled reference's mate-rescue batch (sort_classify + mem_sam_pe_batch with the AVX-512
    kswv kernels; tests/golden/kswv_golden.npz): 5 500 jobs in three sets -- int8 and int16 classes, reads inside / hanging over / missing
    their window, tandem copies (second-best score), N, low-complexity sequence (equal row maxima), other scoring parameters.  All seven
    fields: score, te, qe, score2, te2, tb, qb."""
    import numpy as np
    from common import KSWV_GOLDEN_SETS, kswv_workload
    G = np.load(os.path.join(GOLDEN, "kswv_golden.npz"))
    for name, kw, pen in KSWV_GOLDEN_SETS:
        jobs, ref, qer = kswv_workload(**kw)
        got, cells = O.kswv_batch(jobs, ref, qer, threads=4, **pen)
        got = got.view(np.int32).reshape(-1, 7)
        bad = np.nonzero((got != G[name]).any(axis=1))[0]
        assert bad.size == 0, (name, int(bad[0]), got[bad[0]].tolist(), G[name][bad[0]].tolist())
        assert cells > 0


def test_matesw_pose_oracle_equals_reference_golden():
    """The posing step of mate rescue (orc_matesw_pose: mem_sam_pe_batch_pre + mem_matesw_batch_pre, reference src/bwamem_pair.cpp:660-716, 1060-1223) against
    tests/golden/matesw_golden.npz -- the compiled reference's own function over the same alignment records, worker batch by worker batch: the job index
    array, the jobs' window / query lengths and flags, their sequences (checksums), and the reference's kswv results for them through the oracle's kswv."""
    import zlib
    z = np.load(os.path.join(GOLDEN, "matesw_golden.npz"))
    for tag in ("a", "b"):
        g = z[tag + "_genome"]
        text = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
        n = z[tag + "_read_len"].shape[0]
        for b, first in enumerate(range(0, n, 512)):
            count = min(512, n - first)
            gar, jobs = O.matesw_pose(z[tag + "_regs"], z[tag + "_reg_off"], first, count, z[tag + "_read_len"], z[tag + "_pes"], int(z[tag + "_l_pac"]), z[tag + "_contig_off"], z[tag + "_contig_len"])
            g0, g1 = z[tag + "_gar_off"][b], z[tag + "_gar_off"][b + 1]
            j0, j1 = z[tag + "_job_off"][b], z[tag + "_job_off"][b + 1]
            assert np.array_equal(gar, z[tag + "_gar"][g0:g1]), (tag, b)
            assert jobs.shape[0] == j1 - j0
            assert np.array_equal(jobs["len1"], z[tag + "_len1"][j0:j1]) and np.array_equal(jobs["len2"], z[tag + "_len2"][j0:j1]) and np.array_equal(jobs["xtra"], z[tag + "_xtra"][j0:j1])
            kj, ref, qer = O.matesw_job_seqs(jobs, text, z[tag + "_reads"], z[tag + "_read_off"])
            assert (zlib.crc32(ref.tobytes()), zlib.crc32(qer.tobytes())) == tuple(int(x) for x in z[tag + "_seq_crc"][b]), (tag, b)
            if jobs.shape[0]:
                got, _ = O.kswv_batch(kj, ref, qer)
                assert np.array_equal(got, z[tag + "_kswr"][j0:j1].view(got.dtype).reshape(got.shape) if z[tag + "_kswr"].dtype != got.dtype else z[tag + "_kswr"][j0:j1]), (tag, b)
