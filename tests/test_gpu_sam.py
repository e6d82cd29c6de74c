"""SAM text on the device (meme_sam_format_batch_host = mem_aln2sam for a chunk's plain records), through the C ABI, against the text the compiled
reference's mem_aln2sam wrote for the same records (tests/golden/sam_golden.npz) and against the oracle with other options."""
import os

import numpy as np
import pytest

import oracle_py as O
from common import GOLDEN, build_index, sam_workload
from pymeme import hipapi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def staged(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("sam")
    recs, blob, names, reads, quals, contigs = sam_workload()
    g = synth.make_genome(120_000, seed=5)
    fa = str(tmp / "g.fa")
    synth.write_fasta(fa, g, contigs=2)
    prefix = build_index(fa, bits=12)
    ctx = hipapi.Context(0)
    ctx.load_index_files(prefix)
    off = np.zeros(len(reads) + 1, np.int64)
    off[1:] = np.cumsum([len(r) for r in reads])
    ctx.seed_batch_host(np.concatenate(reads), off)            # stages the reads the records refer to
    yield ctx, recs, blob, names, reads, quals, contigs
    ctx.close()


def test_device_sam_text_equals_reference_golden(staged):
    ctx, recs, blob, names, reads, quals, contigs = staged
    G = np.load(os.path.join(GOLDEN, "sam_golden.npz"))
    # the fixture mixes reads with and without qualities; the device stages qualities per batch: two passes over the two kinds
    have = np.array([q is not None for q in quals])
    for with_q in (True, False):
        sel = np.nonzero(have == with_q)[0]
        qbuf = b"".join(quals[k] if quals[k] is not None else b"!" * len(reads[k]) for k in range(len(reads))) if with_q else None
        ctx.sam_stage_text(names, qbuf)
        for softclip, rg in ((0, b""), (1, b"grp1")):
            want_text, want_off = G["text_%d" % softclip].tobytes(), G["off_%d" % softclip]
            text, off, ms = ctx.sam_format_batch_host(recs[sel], blob, contigs, softclip, rg)
            assert off.shape[0] == sel.shape[0] + 1 and off[0] == 0 and off[-1] == len(text)
            for i, k in enumerate(sel):
                assert text[off[i]:off[i + 1]] == want_text[want_off[k]:want_off[k + 1]], (int(k), softclip, text[off[i]:off[i + 1]], want_text[want_off[k]:want_off[k + 1]])


def test_device_sam_text_errors_and_empty(staged):
    ctx, recs, blob, names, reads, quals, contigs = staged
    ctx.sam_stage_text(names, None)
    text, off, _ = ctx.sam_format_batch_host(recs[:0], blob, contigs)
    assert text == b""
    # empty slots (read = -1) between records: no text for them, the others unchanged
    holes = recs[:6].copy()
    holes["read"][[1, 4]] = -1
    cb0, co0 = O.contig_table(contigs)
    text, off, _ = ctx.sam_format_batch_host(holes, blob, contigs)
    assert off[2] == off[1] and off[5] == off[4]
    for k in (0, 2, 3, 5):
        assert text[off[k]:off[k + 1]] == O.aln2sam(recs[k], blob, names[k], reads[k], None, cb0, co0)
    bad = recs[:4].copy()
    bad["read"][1] = len(reads) + 3
    with pytest.raises(hipapi.MemeError, match="malformed"):
        ctx.sam_format_batch_host(bad, blob, contigs)
    bad = recs[:4].copy()
    k = int(np.nonzero(bad["n_cigar"] > 0)[0][0])
    bad["cigar_off"][k] = blob.shape[0] - 2
    with pytest.raises(hipapi.MemeError, match="malformed"):
        ctx.sam_format_batch_host(bad, blob, contigs)
    # a blob whose last string is not terminated
    cut = recs[-1:].copy()
    if cut["xa_off"][0] < 0:
        cut["xa_off"][0] = 0
    with pytest.raises(hipapi.MemeError, match="does not end inside the blob|malformed"):
        ctx.sam_format_batch_host(cut, np.full(64, 65, np.uint8), contigs)
    # one long name, a one-base read: the oracle's text
    cb, co = O.contig_table(contigs)
    r = recs[:1].copy()
    got, off, _ = ctx.sam_format_batch_host(r, blob, contigs, 0, b"x" * 200)
    assert got == O.aln2sam(r[0], blob, names[0], reads[0], None, cb, co, 0, b"x" * 200)
