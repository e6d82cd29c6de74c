"""Seed extension on the device (meme_extend_last_batch_host = mem_chain2aln_across_reads_V2 behind the chaining kernels), through the
C ABI, against the records the compiled reference made (tests/golden/ext_golden.npz) and against the oracle with other options."""
import os

import numpy as np
import pytest

import oracle_py as O
from common import GOLDEN, build_index, ext_golden_inputs
from pymeme import hipapi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[1, 0], ids=["eight-lanes-per-light-read", "wavefront-per-read"])
def _ext_split(request, monkeypatch):
    """Round 6: the stage's read-walking kernels run eight lanes per read for reads with at most 8 chained seeds (tuning "ext_split", the default) and a
    wavefront per read for the rest -- every test of this file both ways."""
    monkeypatch.setenv("MEME_TUNING", "ext_split=%d" % request.param)


def _device_records(tmp_path, I, ext_opt=None, ascii=False, chain_opt=None, live_only=False, rounds=None):
    fa = str(tmp_path / "c.fa")
    synth.write_fasta(fa, I["genome"], name="cg", contigs=3)
    prefix = build_index(fa, bits=14)
    ctx = hipapi.Context(0)
    try:
        ctx.load_index_files(prefix)
        if live_only:
            ctx.set_tuning("ext_live_only", 1)
        if rounds is not None:
            ctx.set_tuning("ext_rounds", rounds)
        if ascii:       # the reads as FASTQ letters (mixed case, N and other IUPAC letters for ambiguous bases): converted on the device
            letters = np.frombuffer(b"ACGTN", np.uint8)[np.minimum(I["reads"], 4)].copy()
            rng = np.random.default_rng(3)
            low = rng.random(letters.shape[0]) < 0.3
            letters[low] |= 0x20
            amb = I["reads"] >= 4
            letters[amb] = rng.choice(np.frombuffer(b"NnRYK.-", np.uint8), size=int(amb.sum()))
            ctx.seed_batch_resident_ascii(letters, I["read_off"])
        else:
            ctx.seed_batch_resident(I["reads"], I["read_off"])
        contigs = [(int(o), int(l), 0) for o, l in zip(I["contig_off"], I["contig_len"])]
        return ctx.extend_last_batch_host(contigs, chain_opt or hipapi.default_chain_opt(I["l_pac"]), ext_opt)
    finally:
        ctx.close()


def _assert_same(regs, want_cols_or_regs, frac_bits):
    for k, f in enumerate(O.ALNREG_FIELDS):
        want = want_cols_or_regs[:, k] if isinstance(want_cols_or_regs, np.ndarray) and want_cols_or_regs.dtype == np.int64 else want_cols_or_regs[f].astype(np.int64)
        bad = np.nonzero(regs[f].astype(np.int64) != want)[0]
        assert bad.size == 0, (f, int(bad[0]), int(regs[f][bad[0]]), int(want[bad[0]]))
    assert np.array_equal(regs["frac_rep"].view(np.uint32), frac_bits)
    assert not regs["n_comp_is_alt"].any() and not regs["hash"].any() and not regs["flg"].any()


def test_device_records_equal_reference_golden(tmp_path):
    I = ext_golden_inputs()
    G = np.load(os.path.join(GOLDEN, "ext_golden.npz"))
    R = _device_records(tmp_path, I)
    assert np.array_equal(R["reg_off"], G["reg_off"])           # same chains, hence one record per chained seed in the same places
    _assert_same(R["regs"], G["regs"], G["frac_rep_bits"])
    assert R["n_retried"] > 300 and R["n_pairs"] > R["regs"].shape[0]
    purged = (R["regs"]["qb"] == -1) & (R["regs"]["qe"] == -1)
    assert int(purged.sum()) == int(((G["regs"][:, 2] == -1) & (G["regs"][:, 3] == -1)).sum()) > 5000


def _live_of(reg_off, qb, qe):
    """what mem_kernel2_core keeps of the stage's records (src/bwamem.cpp:1680-1693): the mask and the offsets of the kept ones"""
    keep = np.asarray(qe) > np.asarray(qb)
    cs = np.concatenate([[0], np.cumsum(keep.astype(np.int64))])
    return keep, cs[np.asarray(reg_off)]


@pytest.mark.parametrize("rounds", [0, 1, 2, 6])
def test_live_only_hand_over_is_the_golden_records_minus_the_purged_ones(tmp_path, rounds):
    """Tuning "ext_live_only": the compaction mem_kernel2_core runs first on the stage's records (src/bwamem.cpp:1680-1693: qe > qb kept,
    order kept) done on the device -- what the bound aligner asks for, so that purged records do not cross to the host.  With "ext_rounds" > 0
    the stage also stops extending seeds the purge drops (extension in rounds, k_ext_advance): the surviving records are the compiled reference's."""
    I = ext_golden_inputs()
    G = np.load(os.path.join(GOLDEN, "ext_golden.npz"))
    R = _device_records(tmp_path, I, live_only=True, rounds=rounds)
    keep, want_off = _live_of(G["reg_off"], G["regs"][:, 2], G["regs"][:, 3])
    assert 0 < int(keep.sum()) < keep.shape[0] and R["total_seeds"] == keep.shape[0]
    assert np.array_equal(R["reg_off"], want_off)
    _assert_same(R["regs"], G["regs"][keep], G["frac_rep_bits"][keep])
    if rounds == 0:
        assert R["n_ext_seeds"] == keep.shape[0]
    else:
        assert int(keep.sum()) <= R["n_ext_seeds"] < keep.shape[0]          # fewer seeds extended than chained, at least the survivors
    if rounds == 6:
        assert R["n_ext_seeds"] < 0.62 * keep.shape[0]


def test_reads_as_fastq_letters_give_the_same_records(tmp_path):
    """meme_seed_batch_resident_ascii: the base-code conversion of mem_kernel1_core_Learned (src/bwamem.cpp:1277-1279) on the device."""
    I = ext_golden_inputs()
    G = np.load(os.path.join(GOLDEN, "ext_golden.npz"))
    R = _device_records(tmp_path, I, ascii=True)
    assert np.array_equal(R["reg_off"], G["reg_off"])
    _assert_same(R["regs"], G["regs"], G["frac_rep_bits"])


@pytest.mark.parametrize("w,clip,zdrop", [(20, 5, 100), (100, 0, 30)])
def test_device_records_equal_oracle_with_other_options(tmp_path, w, clip, zdrop):
    I = ext_golden_inputs()
    eo = hipapi.default_ext_opt(w)
    eo.pen_clip5 = eo.pen_clip3 = clip
    eo.zdrop = zdrop
    oo = O.default_ext_opt(w)
    oo.pen_clip5 = oo.pen_clip3 = clip
    oo.zdrop = zdrop
    R = _device_records(tmp_path, I, eo)
    want, _ = O.extend_batch(I["reads"], I["read_off"], I["chain_off"], I["chains"], I["seed_off"], I["seeds"], I["frac_rep"], I["text"], I["l_pac"],
                             I["contig_off"], I["contig_len"], oo)
    assert np.array_equal(R["reg_off"], I["seed_off"])
    _assert_same(R["regs"], want, want["frac_rep"].view(np.uint32))
    keep, want_off = _live_of(I["seed_off"], want["qb"], want["qe"])        # ... and in rounds: the survivors only
    R2 = _device_records(tmp_path, I, eo, live_only=True, rounds=2)
    assert np.array_equal(R2["reg_off"], want_off)
    _assert_same(R2["regs"], want[keep], want["frac_rep"].view(np.uint32)[keep])


@pytest.mark.parametrize("W,penalties", [(5, None), (10, None), (20, None), (20, (2, 9, 3, 2, 5, 1))])
def test_seed_filter_on_the_device_equals_oracle(tmp_path, W, penalties):
    """mem_flt_chained_seeds (src/bwamem.cpp:565-598) between chaining and extension.  It runs where 1.1 W <= 0.05 x read length: for every
    read of the batch under -W 5, for the 250-base and longer ones under -W 10, for the noisy 440-500-base reads (LEARNED_MAX_READ_LEN is
    500) under -W 20, where seeds below the bar (22; 44 with match score 2) leave their chains; everywhere the seeds' scores -- the order
    the seeds of a chain are extended in -- become alignment scores.  Records, their number per read and the filter's counters against the
    oracle (pinned on the compiled reference's mem_flt_chained_seeds + extension in tests/test_ref_live.py)."""
    from common import flt_workload
    I = flt_workload(W, lo=420, hi=500)
    co = hipapi.default_chain_opt(I["l_pac"])
    co.min_chain_weight = W
    eo, oo = hipapi.default_ext_opt(), O.default_ext_opt()
    if penalties:
        for o in (eo, oo):
            o.a, o.b, o.o_del, o.e_del, o.o_ins, o.e_ins = penalties
    R = _device_records(tmp_path, I, eo, chain_opt=co)
    chains, soff, seeds, score, n_sw = O.flt_batch(I["reads"], I["read_off"], I["chain_off"], I["chains"], I["seed_off"], I["seeds"], I["text"], I["l_pac"],
                                                   I["contig_off"], I["contig_len"], oo, W)
    want, _ = O.extend_batch(I["reads"], I["read_off"], I["chain_off"], chains, soff, seeds, I["frac_rep"], I["text"], I["l_pac"], I["contig_off"], I["contig_len"],
                             oo, seed_score=score)
    assert np.array_equal(R["reg_off"], soff)
    _assert_same(R["regs"], want, want["frac_rep"].view(np.uint32))
    assert R["n_flt_jobs"] == n_sw > 500 and R["n_flt_dropped"] == int(I["seed_off"][-1] - soff[-1])
    assert W < 20 or R["n_flt_dropped"] > 0
    keep, want_off = _live_of(soff, want["qb"], want["qe"])                 # the same behind the filter in rounds (seed scores decide the order)
    R2 = _device_records(tmp_path, I, eo, chain_opt=co, live_only=True, rounds=1)
    assert np.array_equal(R2["reg_off"], want_off) and R2["n_flt_dropped"] == R["n_flt_dropped"] and R2["total_seeds"] == int(soff[-1])
    _assert_same(R2["regs"], want[keep], want["frac_rep"].view(np.uint32)[keep])


def test_seed_filter_is_off_for_short_reads(tmp_path):
    I = ext_golden_inputs()
    R = _device_records(tmp_path, I)
    assert R["n_flt_jobs"] == 0 and R["n_flt_dropped"] == 0


def test_extend_call_needs_a_seeded_batch_and_the_right_genome(tmp_path):
    g = synth.make_genome(60_000, seed=9)
    fa = str(tmp_path / "e.fa")
    synth.write_fasta(fa, g, contigs=1)
    prefix = build_index(fa, bits=12)
    ctx = hipapi.Context(0)
    try:
        ctx.load_index_files(prefix)
        with pytest.raises(hipapi.MemeError, match="no seeded batch"):
            ctx.extend_last_batch_host([(0, 60_000, 0)], hipapi.default_chain_opt(60_000))
        r, _, _ = synth.make_reads(g, 50, 100, seed=10)
        ctx.seed_batch_host(r.reshape(-1), np.arange(0, 51 * 100, 100, dtype=np.int64))
        with pytest.raises(hipapi.MemeError, match="l_pac does not match"):
            ctx.extend_last_batch_host([(0, 50_000, 0)], hipapi.default_chain_opt(50_000))
        res = ctx.extend_last_batch_host([(0, 60_000, 0)], hipapi.default_chain_opt(60_000))
        assert res["reg_off"].shape[0] == 51 and res["reg_off"][-1] == res["regs"].shape[0] >= 40
        a = res["regs"]
        live = ~((a["qb"] == -1) & (a["qe"] == -1))
        assert np.all(a["qe"][live] > a["qb"][live]) and np.all(a["re"][live] > a["rb"][live]) and np.all(a["score"][live] >= 19)
    finally:
        ctx.close()


def test_extension_refuses_seeds_without_their_reads(tmp_path):
    """meme_chain_batch_host brings seeds from the caller and leaves whatever bases an earlier seeding call staged: the calls that read the
    batch's bases (extension, global alignment) must refuse instead of working on stale bytes."""
    I = ext_golden_inputs()
    fa = str(tmp_path / "c.fa")
    synth.write_fasta(fa, I["genome"], name="cg", contigs=3)
    prefix = build_index(fa, bits=14)
    ctx = hipapi.Context(0)
    try:
        ctx.load_index_files(prefix)
        smems, smem_off, hits, hit_off = ctx.seed_batch_host(I["reads"], I["read_off"])
        contigs = [(int(o), int(l), 0) for o, l in zip(I["contig_off"], I["contig_len"])]
        read_len = np.diff(I["read_off"]).astype(np.int32)
        ctx.chain_batch_host(smems, smem_off, hits, hit_off, read_len, contigs, hipapi.default_chain_opt(I["l_pac"]))
        with pytest.raises(hipapi.MemeError, match="no seeded batch"):
            ctx.extend_last_batch_host(contigs, hipapi.default_chain_opt(I["l_pac"]), None)
        ctx.seed_batch_resident(I["reads"], I["read_off"])          # a seeding call makes the ctx whole again
        R = ctx.extend_last_batch_host(contigs, hipapi.default_chain_opt(I["l_pac"]), None)
        assert R["regs"].shape[0] > 0
    finally:
        ctx.close()
