"""Live differential tests against the COMPILED REFERENCE (oracle/_ref, built by oracle/Makefile.ref where /root/reference exists).
Skipped where the reference binaries are absent or cannot run on the host CPU.  They pin the oracle's restatements on inputs the
committed fixtures cannot hold in bulk."""
import os

import numpy as np
import pytest

import chain_gen
import oracle_py as O
import ref_py

needs_stage = pytest.mark.skipif(not (ref_py.have("libstage_ref.so") and ref_py.cpu_can_run()), reason="compiled reference (libstage_ref.so) not available")


def _same_chains(a, b):
    rc_a, ch_a, sd_a, tree_a, frac_a = a
    rc_b, ch_b, sd_b, tree_b, frac_b = b
    if rc_a != rc_b or tree_a != tree_b:
        return False
    if rc_a > 0 and np.float32(frac_a).view(np.uint32) != np.float32(frac_b).view(np.uint32):
        return False
    for f in ("pos", "rid", "n_seeds", "w", "first", "kept", "is_alt", "seed_beg"):
        if not np.array_equal(ch_a[f], ch_b[f]):
            return False
    return all(np.array_equal(sd_a[f], sd_b[f]) for f in ("rbeg", "qbeg", "len"))


@needs_stage
def test_chain_oracle_equals_reference_on_adversarial_reads():
    """orc_chain_read == mem_chain_Learned + mem_chain_flt of the compiled reference on 1 500 made-up reads with up to hundreds of
    chains, most of them with several chains at EQUAL positions (the B-tree's placement of equal keys decides their order)."""
    l_pac = 200_000
    contig_off = np.array([0, 70_000, 150_000], np.int64)
    contig_len = np.array([70_000, 80_000, 50_000], np.int32)
    alt = np.array([0, 0, 1], np.uint8)
    opt = O.default_chain_opt(l_pac)
    n_dup = n_deep = 0
    for r, (sm, hits) in enumerate(chain_gen.workload(77, 1500, l_pac=l_pac)):
        L = 250 if r % 5 == 2 else 150
        want = ref_py.chain_read(sm, hits, L, contig_off, contig_len, alt, opt)
        got = O.chain_read(sm, hits, L, contig_off, alt, opt, chain_cap=8192, seed_cap=1 << 17)
        assert _same_chains(got, want), (r, got[0], want[0], got[3], want[3])
        n_deep += want[3] > 9
        n_dup += len(np.unique(want[1]["pos"])) < want[0]
    assert n_dup > 300 and n_deep > 300, (n_dup, n_deep)


@needs_stage
@pytest.mark.parametrize("change", [dict(w=3), dict(w=2000), dict(max_chain_gap=40), dict(max_occ=37), dict(drop_ratio=0.9, mask_level=0.1),
                                    dict(max_chain_extend=3), dict(min_chain_weight=40)])
def test_chain_oracle_equals_reference_under_other_options(change):
    """The option sets tests/test_gpu_chain.py runs the device under, oracle against the compiled reference (band, gap limit, hit sampling,
    filter thresholds, the cap on extended chains, the weight floor).  With a weight floor the reference has a quirk the oracle restates:
    when ALL chains of a read are below it, mem_chain_flt returns the stale first element of its array (src/bwamem.cpp:604-643) -- and
    aborts in free() later when that chain had grown beyond SEEDS_PER_CHAIN (= 1) seeds, which it frees before returning it; those reads
    are left out of the comparison."""
    l_pac = 200_000
    contig_off = np.array([0, 70_000, 150_000], np.int64)
    contig_len = np.array([70_000, 80_000, 50_000], np.int32)
    alt = np.array([0, 0, 1], np.uint8)
    opt = O.default_chain_opt(l_pac)
    for k, v in change.items():
        setattr(opt, k, v)
    n_stale = 0
    for r, (sm, hits) in enumerate(chain_gen.workload(78, 400, l_pac=l_pac)):
        L = 250 if r % 5 == 2 else 150
        got = O.chain_read(sm, hits, L, contig_off, alt, opt, chain_cap=8192, seed_cap=1 << 17)
        if "min_chain_weight" in change:
            stale = got[0] == 1 and int(got[1]["w"][0]) < opt.min_chain_weight
            n_stale += stale
            if stale and int(got[1]["n_seeds"][0]) > 1:           # (SEEDS_PER_CHAIN = 1: the reference has freed that chain's seeds before it returns it)
                continue
        want = ref_py.chain_read(sm, hits, L, contig_off, contig_len, alt, opt)
        assert _same_chains(got, want), (change, r, got[0], want[0], got[3], want[3])
    assert "min_chain_weight" not in change or n_stale > 5


@needs_stage
@pytest.mark.parametrize("w,clip,zdrop", [(100, 5, 100), (20, 5, 100), (100, 0, 30), (7, 11, 100)])
def test_ext_oracle_equals_reference_on_other_options(w, clip, zdrop):
    """orc_extend_batch == the compiled reference's mem_chain2aln_across_reads_V2 on the extension fixture's reads and chains with
    other band widths (narrow bands: most jobs retried), clipping penalties and z-drop than the committed fixture was made with."""
    from common import ext_golden_inputs
    I = ext_golden_inputs()
    opt = O.default_ext_opt(w)
    opt.pen_clip5 = opt.pen_clip3 = clip
    opt.zdrop = zdrop
    sel = slice(0, None)
    want = ref_py.extend_reads(I["reads"], I["read_off"], I["chain_off"], I["chains"], I["seed_off"], I["seeds"], I["frac_rep"], I["text"], I["l_pac"],
                               I["contig_off"], I["contig_len"], np.zeros(I["contig_off"].shape[0], np.uint8), opt)
    got, (jobs, retried) = O.extend_batch(I["reads"], I["read_off"], I["chain_off"], I["chains"], I["seed_off"], I["seeds"], I["frac_rep"], I["text"],
                                          I["l_pac"], I["contig_off"], I["contig_len"], opt)
    for f in O.ALNREG_FIELDS:
        bad = np.nonzero(got[f][sel] != want[f][sel])[0]
        assert bad.size == 0, (f, int(bad[0]), int(got[f][bad[0]]), int(want[f][bad[0]]))
    assert np.array_equal(got["frac_rep"].view(np.uint32), want["frac_rep"].view(np.uint32))


@needs_stage
@pytest.mark.parametrize("W,penalties,span", [(0, None, (800, 1300)), (5, None, (800, 1300)), (10, None, (800, 1300)), (0, (2, 9, 3, 2, 5, 1), (800, 1300)),
                                              (20, None, (420, 500)), (20, (2, 9, 3, 2, 5, 1), (420, 500))])
def test_seed_filter_oracle_equals_reference(W, penalties, span):
    """orc_flt_batch == the compiled reference's mem_flt_chained_seeds (mem_seed_sw -> ksw_align2 -> ksw_i16) on long noisy reads (the
    filter runs without -W for reads the function takes although the learned-index path stops at 500 bases), on 150 / 250-base reads under
    -W 5 / -W 10 (at 10 it runs for the 250-base reads of the batch only) and on 420-500-base reads under -W 20 (the GPU tests' workload);
    then the extension with the scores the filter leaves, oracle against reference.  Other penalties: match 2, mismatch 9, asymmetric gaps."""
    from common import flt_workload
    I = flt_workload(W, lo=span[0], hi=span[1])
    opt = O.default_ext_opt()
    if penalties:
        opt.a, opt.b, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins = penalties
    alt = np.zeros(I["contig_off"].shape[0], np.uint8)
    want = ref_py.flt_chained_seeds(I["reads"], I["read_off"], I["chain_off"], I["chains"], I["seed_off"], I["seeds"], I["text"], I["l_pac"], I["contig_off"],
                                    I["contig_len"], alt, opt, W)
    got = O.flt_batch(I["reads"], I["read_off"], I["chain_off"], I["chains"], I["seed_off"], I["seeds"], I["text"], I["l_pac"], I["contig_off"], I["contig_len"],
                      opt, W)
    assert got[4] > 500                                            # alignments run
    assert np.array_equal(got[1], want[1])
    assert W in (5, 10) or 0 < int(got[1][-1]) < int(I["seed_off"][-1])  # some seeds left their chains (under -W 5 / 10 the bar is 5 / 11: below any seed's own score)
    for f in ("seed_beg", "n_seeds"):
        assert np.array_equal(got[0][f], want[0][f]), f
    for f in ("rbeg", "qbeg", "len"):
        assert np.array_equal(got[2][f], want[2][f]), f
    assert np.array_equal(got[3], want[3])
    # (the windows the aligner's call reads are scrambled -- see orc_seed_sw -- so an alignment scores what two unrelated sequences score:
    # above the bar of -W 5 / 10, where such a seed stays with that score, below the others, where it leaves)
    assert W not in (5, 10) or (got[3] != got[2]["len"] * opt.a).any()
    r_want = ref_py.extend_reads(I["reads"], I["read_off"], I["chain_off"], want[0], want[1], want[2], I["frac_rep"], I["text"], I["l_pac"], I["contig_off"],
                                 I["contig_len"], alt, opt, seed_score=want[3])
    r_got, _ = O.extend_batch(I["reads"], I["read_off"], I["chain_off"], got[0], got[1], got[2], I["frac_rep"], I["text"], I["l_pac"], I["contig_off"],
                              I["contig_len"], opt, seed_score=got[3])
    for f in O.ALNREG_FIELDS:
        bad = np.nonzero(r_got[f] != r_want[f])[0]
        assert bad.size == 0, (f, int(bad[0]), int(r_got[f][bad[0]]), int(r_want[f][bad[0]]))


@needs_stage
def test_ksw_global2_oracle_equals_reference_on_other_penalties():
    """orc_ksw_global2 == the compiled reference's ksw_global2 with other scores (asymmetric gap penalties, cheap gaps: the case the
    reference's comment warns about, insertion next to deletion) on a sample of the fixture's alignments."""
    from common import gcig_workload
    _, _, jobs, seqs = gcig_workload(n=600, seed=91)
    for a, b, od, ed, oi, ei in ((1, 4, 6, 1, 6, 1), (2, 3, 4, 2, 7, 1), (1, 9, 1, 1, 1, 1), (3, 1, 5, 3, 2, 2)):
        for J, (q, t) in zip(jobs, seqs):
            want = ref_py.ksw_global2(q, t, int(J["w"]), a, b, od, ed, oi, ei)
            got = O.ksw_global2(q, t, int(J["w"]), a, b, od, ed, oi, ei)
            assert got[0] == want[0] and np.array_equal(got[1], want[1]), (a, b, od, ed, oi, ei, int(J["w"]), got[0], want[0])


@needs_stage
def test_gen_cigar2_oracle_equals_reference_on_other_penalties():
    """orc_gen_cigar2 == the compiled reference's bwa_gen_cigar2 (score, CIGAR, NM, MD) with other scores: the band derivation
    (src/bwa.cpp:306-316) depends on match score and gap penalties."""
    from common import gencig_workload
    from pymeme import hipapi
    g, reads, calls = gencig_workload(n=500, seed=211)
    text = hipapi.fwd_rc_text(g)
    for pen in ((2, 3, 4, 2, 7, 1), (1, 9, 1, 1, 1, 1), (3, 1, 5, 3, 2, 2)):
        for J in calls:
            q = reads[int(J["read"])][int(J["qb"]):int(J["qb"]) + int(J["qlen"])]
            rb, re = int(J["rb"]), int(J["rb"]) + int(J["tlen"])
            want = ref_py.gen_cigar2(g, q, rb, re, int(J["w_"]), *pen)
            got = O.gen_cigar2(text, g.shape[0], q, rb, re, int(J["w_"]), *pen)
            assert got[0] == want[0] and np.array_equal(got[1], want[1]) and got[2:] == want[2:], (pen, J, got[0], want[0], got[2:], want[2:])
    assert ref_py.gen_cigar2(g, reads[0][:50], g.shape[0] - 20, g.shape[0] + 30, 10) is None


@needs_stage
def test_kswv_oracle_equals_reference_live():
    """orc_kswv_batch == the compiled reference's mate-rescue batch (ref_kswv_batch: sort_classify + mem_sam_pe_batch, AVX-512 kswv kernels) on
    fresh job sets: other seeds, short reads, extreme penalties (free gap opens; mismatch 9), windows shorter than the read."""
    from common import kswv_workload
    if ref_py.stage_lib().ref_kswv_batch(None, 0, None, 0, None, 0, 1, 4, 6, 1, 6, 1, None) != 0:
        pytest.skip("the compiled reference is not an AVX-512 build (no batched kswv kernels)")
    for kw, pen in ((dict(n=1200, seed=17), {}), (dict(n=800, seed=18, read_len=(19, 140)), dict(a=1, b=9, o_del=1, e_del=1, o_ins=1, e_ins=1)),
                    (dict(n=800, seed=19, read_len=(240, 260)), dict(a=1, b=1, o_del=0, e_del=1, o_ins=0, e_ins=1)),
                    (dict(n=600, seed=20, read_len=(100, 500), a=3), dict(a=3, b=5, o_del=7, e_del=2, o_ins=3, e_ins=3))):
        jobs, ref, qer = kswv_workload(**kw)
        want = ref_py.kswv_batch(jobs, ref, qer, **pen).view(np.int32).reshape(-1, 7)
        got = O.kswv_batch(jobs, ref, qer, threads=4, **pen)[0].view(np.int32).reshape(-1, 7)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, (kw, pen, int(bad[0]), got[bad[0]].tolist(), want[bad[0]].tolist())


@needs_stage
def test_kswv_edge_jobs_reference_live():
    """The ten edge jobs of tests/common.py kswv_edge_jobs(): compiled reference == oracle == the records kept in KSWV_EDGE_WANT."""
    from common import KSWV_EDGE_WANT, kswv_edge_jobs
    if ref_py.stage_lib().ref_kswv_batch(None, 0, None, 0, None, 0, 1, 4, 6, 1, 6, 1, None) != 0:
        pytest.skip("the compiled reference is not an AVX-512 build (no batched kswv kernels)")
    jobs, rb, qb = kswv_edge_jobs()
    want = ref_py.kswv_batch(jobs, rb, qb).view(np.int32).reshape(-1, 7)
    assert np.array_equal(want, np.array(KSWV_EDGE_WANT, np.int32))
    assert np.array_equal(O.kswv_batch(jobs, rb, qb)[0].view(np.int32).reshape(-1, 7), want)


needs_aligner = pytest.mark.skipif(not (ref_py.have("bwa-meme_mode3") and ref_py.cpu_can_run()), reason="compiled reference (bwa-meme_mode3) not available")


@needs_aligner
@pytest.mark.parametrize("case", ["plain", "runs_and_N"])
def test_index_files_byte_identical_to_bwa_meme_index(tmp_path, case):
    """`meme-index build` (bwa-meme_amd/host) against `bwa-meme index -a meme` (reference src/Learnedindex.cpp:456-548, src/bwtindex.cpp,
    src/bntseq.cpp:313-371) on the same FASTA: all six files the learned path reads -- .pac .ann .amb .0123 .pos_packed
    .suffixarray_uint64 -- byte for byte.  Second case: several contigs, long A / T runs at contig ends (the T-padding rule of the
    suffix order) and ambiguous bases (replaced by the reference's srand48(11) stream, recorded in .amb)."""
    import filecmp
    import shutil
    import subprocess
    from pymeme import synth
    rng = np.random.default_rng(5)
    if case == "plain":
        g = synth.make_genome(200_000, seed=31, repeat_frac=0.05)
        fa = str(tmp_path / "a.fa")
        synth.write_fasta(fa, g, contigs=2)
    else:
        g = synth.make_genome(120_000, seed=32, repeat_frac=0.03)
        g[:40] = 0; g[-55:] = 3; g[60_000:60_070] = 3; g[30_000:30_033] = 0
        fa = str(tmp_path / "a.fa")
        seq = np.frombuffer(b"ACGT", np.uint8)[g].copy()
        for p in rng.integers(100, g.shape[0] - 100, size=12):
            seq[p:p + int(rng.integers(1, 30))] = ord("N")
        cuts = [0, 25_000, 60_035, 90_001, g.shape[0]]
        with open(fa, "w") as fh:
            for k in range(4):
                fh.write(">ctg%d some text\n" % k)
                s = seq[cuts[k]:cuts[k + 1]].tobytes().decode()
                for i in range(0, len(s), 70):
                    fh.write(s[i:i + 70] + "\n")
    ours = str(tmp_path / "ours.fa")
    theirs = str(tmp_path / "theirs.fa")
    shutil.copy(fa, ours)
    shutil.copy(fa, theirs)
    subprocess.run([os.path.join(O.REPO, "bwa-meme_amd", "meme-index"), "build", ours, "-b", "12", "-t", "4"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(ref_py.REF_DIR, "bwa-meme_mode3"), "index", "-a", "meme", "-t", "4", theirs], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    for ext in (".pac", ".ann", ".amb", ".0123", ".pos_packed", ".suffixarray_uint64"):
        assert os.path.exists(theirs + ext), ext
        if ext == ".ann":      # (the first line carries the FASTA's seed-independent numbers only; compare as text)
            assert open(ours + ext).read() == open(theirs + ext).read(), ext
        else:
            assert filecmp.cmp(ours + ext, theirs + ext, shallow=False), ext


@pytest.mark.skipif(not (ref_py.have("learned_seeding_count1") and ref_py.have("learned_seeding_count3") and ref_py.cpu_can_run()),
                    reason="compiled reference with Count_mem_ref (oracle/Makefile.ref) not available")
def test_refpath_work_counters_match_the_reference_own_counters(tmp_path):
    """oracle/meme_refpath.c's work counters (what bench.py's roofline numerator is made of: model lookups and suffix-array compares per
    read) against the REFERENCE'S OWN counters: the seeding harness rebuilt with Count_mem_ref 1 (src/LearnedIndex_seeding.h:94) prints,
    per search, how many entries it compared (binary + linear + min_intv phases).  MODE 1 has the probe sequence the restatement
    follows (no inverse suffix array).  A few of the reference's exits return without printing their line (~4 % of the searches), so it
    reports slightly fewer searches AND leaves out those searches' compares: searches must agree within 6 %, compares within 3 %, and
    the surplus of compares must be what the unprinted searches account for (a handful of compares each).  MODE 3 (ISA shortcuts for reads that matched end to end) does less work: reported."""
    import re
    import subprocess
    from common import build_index
    from pymeme import synth
    g = synth.make_genome(2_000_000, seed=301, repeat_frac=0.04)
    fa = str(tmp_path / "c.fa")
    synth.write_fasta(fa, g, contigs=2)
    prefix = build_index(fa, bits=16)
    reads, _, _ = synth.make_reads(g, 3000, 150, seed=302, n_frac=0.02)
    fq = str(tmp_path / "c.fq")
    synth.write_fastq(fq, reads, prefix="c")
    idx = O.load_index_files(prefix)
    l1, l2 = O.load_prmi_files(prefix)
    off = np.arange(0, (reads.shape[0] + 1) * 150, 150, dtype=np.int64)
    _, _, ctr = O.refpath_seed_batch(idx, l1, l2, reads.reshape(-1), off, threads=1)
    got = {}
    for mode in (1, 3):
        r = subprocess.run([os.path.join(ref_py.REF_DIR, "learned_seeding_count%d" % mode), prefix, fq, "1000", "1", "3"], capture_output=True, text=True)
        assert r.returncode == 0
        lines = compares = 0
        for line in r.stdout.splitlines():
            m = re.search(r"Count Total: (\d+)", line)
            if m:
                lines += 1; compares += int(m.group(1))
                continue
            m = re.search(r"Count_bs:(\d+) Count_linear:(\d+) Count_minintv:(\d+)", line)
            if m:
                lines += 1; compares += sum(int(x) for x in m.groups())
        got[mode] = (lines, compares)
    lines1, cmp1 = got[1]
    assert lines1 <= ctr["lookups"] <= 1.06 * lines1, (ctr, got)
    assert cmp1 <= ctr["compares"] <= 1.03 * cmp1, (ctr, got)
    assert ctr["compares"] - cmp1 <= 8 * (ctr["lookups"] - lines1), (ctr, got)
    assert got[3][1] < cmp1                         # the ISA shortcut saves compares, not searches
    print("MODE 3 / MODE 1 compares: %.3f; restatement / MODE 1: %.4f" % (got[3][1] / cmp1, ctr["compares"] / cmp1))


@pytest.mark.skipif(not (ref_py.have("libstage_ref.so") and ref_py.cpu_can_run()), reason="compiled reference (oracle/_ref) not available on this box")
def test_matesw_pose_oracle_equals_reference_live():
    """orc_matesw_pose against the compiled reference's mem_sam_pe_batch_pre on other workloads than the fixture's: every orientation allowed, none allowed,
    narrow and wide insert-size bounds, short reads (windows below min_seed_len), other max_matesw / pen_unpaired / min_seed_len."""
    import oracle_py as O
    from common import matesw_pose_workload
    cases = [(311, [(10, 2000, 0)] * 4, {}), (312, [(0, 0, 1)] * 4, {}), (313, [(0, 0, 1), (200, 260, 0), (0, 0, 1), (150, 151, 0)], {}),
             (314, [(50, 500, 0), (120, 680, 0), (100, 900, 0), (0, 0, 1)], dict(max_matesw=2, pen_unpaired=40)),
             (315, None, dict(min_seed_len=40, a=2)), (316, [(0, 30, 0), (0, 25, 0), (0, 0, 1), (5, 35, 0)], dict(read_len=(20, 60)))]
    for seed, pes, kw in cases:
        rl = kw.pop("read_len", (100, 251))
        W = matesw_pose_workload(seed=seed, pes=pes, n_pairs=300, read_len=rl)
        n = W["read_len"].shape[0]
        for first in range(0, n, 512):
            count = min(512, n - first)
            g1, j1 = O.matesw_pose(W["regs"], W["reg_off"], first, count, W["read_len"], W["pes"], W["l_pac"], W["contig_off"], W["contig_len"], **kw)
            g2, j2, ref2, qer2 = ref_py.matesw_pose(W["genome"], W["l_pac"], W["contig_off"], W["contig_len"], W["reads"], W["read_off"], first, count, W["regs"], W["reg_off"], W["pes"], **kw)
            assert np.array_equal(g1, g2), (seed, first)
            kj, ref1, qer1 = O.matesw_job_seqs(j1, W["text"], W["reads"], W["read_off"])
            for f in ("len1", "len2", "xtra", "idr", "idq"):
                assert np.array_equal(kj[f], j2[f]), (seed, f)
            assert np.array_equal(ref1, ref2) and np.array_equal(qer1, qer2), seed
