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
