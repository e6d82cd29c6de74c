"""Parity of the HIP seeding path (through the C ABI) against the oracle and the committed golden dumps."""
import os

import numpy as np
import pytest

import oracle_py as O
from common import GOLDEN, build_index, read_fastq_codes
from pymeme import hipapi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hipapi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def g1(ctx):
    prefix = build_index(os.path.join(GOLDEN, "g1.fa"))
    ctx.load_index_files(prefix)
    return prefix


def _gpu_dump(ctx, reads, off, rounds=3):
    smems, smem_off, hits, hit_off = ctx.seed_batch(reads, off, hipapi.default_seed_opt(rounds=rounds))
    slots, counts, hl = hipapi.smems_to_slots(smems, smem_off, hits, hit_off)
    return O.format_seed_dump(slots, counts, hl)


@pytest.mark.parametrize("length", [150, 250, 60, 25])
def test_hip_seeds_equal_reference_golden(ctx, g1, length):
    reads, off = read_fastq_codes(os.path.join(GOLDEN, "g1_reads_%d.fq" % length))
    want = open(os.path.join(GOLDEN, "g1_seeds_%d.txt" % length)).read()
    assert _gpu_dump(ctx, reads, off) == want


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16, 32])
def test_hip_seeds_equal_golden_for_every_group_width(g1, lanes):
    c = hipapi.Context(0)
    try:
        c.load_index_files(g1)
        c.set_tuning("group_lanes", lanes)
        for length in (150, 250, 60, 25):
            reads, off = read_fastq_codes(os.path.join(GOLDEN, "g1_reads_%d.fq" % length))
            want = open(os.path.join(GOLDEN, "g1_seeds_%d.txt" % length)).read()
            assert _gpu_dump(c, reads, off) == want
    finally:
        c.close()


def test_overflow_tiers_give_the_same_seeds(g1):
    """With 8 SMEM slots per read in the first pass most 150/250-bp reads overflow and are re-run in the bigger tiers
    (the pending / overflow-list ping-pong, slot locators of several tiers in one gather, the 32-lanes-per-read
    instantiation with its 512-entry LDS ring): same dump."""
    c = hipapi.Context(0)
    try:
        c.load_index_files(g1)
        c.set_tuning("smem_cap", 8)
        for length in (150, 250):
            reads, off = read_fastq_codes(os.path.join(GOLDEN, "g1_reads_%d.fq" % length))
            want = open(os.path.join(GOLDEN, "g1_seeds_%d.txt" % length)).read()
            assert _gpu_dump(c, reads, off) == want
            assert c.timings().seed_launches >= 2
    finally:
        c.close()


def test_host_result_variant_equals_capacity_variant(ctx, g1):
    reads, off = read_fastq_codes(os.path.join(GOLDEN, "g1_reads_150.fq"))
    a = ctx.seed_batch(reads, off)
    b = ctx.seed_batch_host(reads, off)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_read_longer_than_500_bases_is_rejected(ctx, g1):
    """The reference exits on such reads (src/bwamem.cpp:1259-1262); the backend fails the call, loudly."""
    reads = np.zeros(150 + 501, np.uint8)
    off = np.array([0, 150, 651], np.int64)
    with pytest.raises(hipapi.MemeError, match="exceeds the learned-index limit"):
        ctx.seed_batch(reads, off)


@pytest.mark.parametrize("rounds", [1, 2, 3])
def test_hip_seeds_equal_oracle_each_round(ctx, g1, rounds):
    idx = O.load_index_files(g1)
    for length in (150, 60):
        reads, off = read_fastq_codes(os.path.join(GOLDEN, "g1_reads_%d.fq" % length))
        sm, ns, hits, nh, _ = O.seed_batch(idx, reads, off, O.default_seed_params(steps=rounds), smem_cap=256,
                                           hit_cap=4096, threads=4)
        assert _gpu_dump(ctx, reads, off, rounds) == O.format_seed_dump(sm, ns, hits)


def _synthetic_case(tmp, n_bases, seed, **kw):
    g = synth.make_genome(n_bases, seed=seed, **kw)
    fa = os.path.join(tmp, "s%d.fa" % seed)
    synth.write_fasta(fa, g, contigs=2)
    return g, build_index(fa, bits=14)


def test_hip_seeds_equal_oracle_repetitive_genome(tmp_path):
    """Bigger, repeat-rich genome: large hit intervals (galloping edges), re-seeding, third round."""
    g, prefix = _synthetic_case(str(tmp_path), 600_000, 31, repeat_frac=0.2, repeat_len=300, n_families=4,
                                divergence=0.02, n_dups=10, dup_len=3000, poly_runs=8)
    idx = O.load_index_files(prefix)
    c = hipapi.Context(0)
    try:
        c.load_index_files(prefix)
        for L, kw in ((150, dict(exact_frac=0.3, n_frac=0.05)), (250, dict(sub_rate=0.05, indel_rate=0.0075)),
                      (40, dict(n_frac=0.3))):
            reads, _, _ = synth.make_reads(g, 3000, L, seed=100 + L, **kw)
            off = np.arange(0, (reads.shape[0] + 1) * L, L, dtype=np.int64)
            sm, ns, hits, nh, _ = O.seed_batch(idx, reads, off, smem_cap=256, hit_cap=1 << 14, threads=0)
            want = O.format_seed_dump(sm, ns, hits)
            assert _gpu_dump(c, reads, off) == want
    finally:
        c.close()


def test_edge_cases(ctx, g1):
    idx = O.load_index_files(g1)
    rng = np.random.default_rng(7)
    text = idx.text
    n = text.shape[0]
    reads = []
    reads.append(np.full(150, 4, np.uint8))                      # all N
    reads.append(text[:150].copy())                              # text start
    reads.append(text[n - 150:].copy())                          # text end (suffixes running off the text)
    reads.append(text[n // 2 - 75:n // 2 + 75].copy())           # fwd/rc junction
    reads.append(np.zeros(150, np.uint8))                        # poly-A
    reads.append(np.full(150, 3, np.uint8))                      # poly-T (sorts at the T-padding end)
    reads.append(rng.integers(0, 4, 150).astype(np.uint8))       # unrelated
    reads.append(text[1000:1019].copy())                         # exactly min_seed_len
    reads.append(text[1000:1018].copy())                         # shorter than min_seed_len
    reads.append(text[2000:2001].copy())                         # 1 base
    r = text[3000:3150].copy(); r[::20] = 4; reads.append(r)     # N every 20 bases
    r = text[5000:5500].copy(); reads.append(r)                  # 500 bases (maximum)
    off = np.zeros(len(reads) + 1, np.int64)
    off[1:] = np.cumsum([x.shape[0] for x in reads])
    flat = np.concatenate(reads)
    sm, ns, hits, nh, _ = O.seed_batch(idx, flat, off, smem_cap=8192, hit_cap=1 << 15, threads=2)
    assert _gpu_dump(ctx, flat, off) == O.format_seed_dump(sm, ns, hits)


def test_empty_batch(ctx, g1):
    smems, smem_off, hits, hit_off = ctx.seed_batch(np.zeros(0, np.uint8), np.zeros(1, np.int64))
    assert smems.shape[0] == 0 and hits.shape[0] == 0 and smem_off.tolist() == [0]


def test_round_trip_property_large(ctx, g1):
    """Size-independent properties on a larger batch: every SMEM really occurs at every reported hit,
    hits of one SMEM are distinct, and results do not depend on batch composition."""
    idx = O.load_index_files(g1)
    g = idx.text[:idx.text.shape[0] // 2]
    reads, _, _ = synth.make_reads(g, 20000, 150, seed=77)
    off = np.arange(0, (reads.shape[0] + 1) * 150, 150, dtype=np.int64)
    smems, smem_off, hits, hit_off = ctx.seed_batch(reads, off)
    text = idx.text
    rng = np.random.default_rng(1)
    for r in rng.integers(0, reads.shape[0], 300):
        for m in smems[smem_off[r]:smem_off[r + 1]]:
            seg = reads[r, m["start"]:m["end"]]
            hv = hits[hit_off[r] + m["hitbeg"]: hit_off[r] + m["hitbeg"] + m["hitcount"]]
            assert len(set(hv.tolist())) == hv.shape[0]
            for h in hv[:4]:
                assert np.array_equal(text[int(h):int(h) + seg.shape[0]], seg)
    # same reads in a different batch order give the same per-read answers
    perm = rng.permutation(2000)
    sub = reads[:2000][perm]
    off2 = np.arange(0, 2001 * 150, 150, dtype=np.int64)
    s2, so2, h2, ho2 = ctx.seed_batch(sub, off2)
    for k in range(0, 2000, 97):
        r = int(perm[k])
        a = smems[smem_off[r]:smem_off[r + 1]]
        b = s2[so2[k]:so2[k + 1]]
        assert np.array_equal(a, b)
        assert np.array_equal(hits[hit_off[r]:hit_off[r + 1]], h2[ho2[k]:ho2[k + 1]])


def test_hip_seeds_equal_oracle_midsize_default_model(tmp_path):
    """16 Mbp repeat-rich genome with the trainer's default leaf count (partial third layer in use), 6 000 reads of three
    lengths: the model, window and gallop paths at a size where predictions are no longer near-exact."""
    from pymeme import workload
    g = synth.make_genome(16_000_000, seed=91, repeat_frac=0.05, repeat_len=400, n_families=8, divergence=0.03,
                          n_dups=20, dup_len=5000, poly_runs=6)
    prefix = workload.build_index_on_disk(g, str(tmp_path), bits=0)
    idx = O.load_index_files(prefix)
    c = hipapi.Context(0)
    try:
        c.load_index_files(prefix)
        for L, kw in ((150, dict(n_frac=0.02)), (250, dict(sub_rate=0.05, indel_rate=0.0075)), (76, dict(exact_frac=0.5))):
            reads, _, _ = synth.make_reads(g, 2000, L, seed=300 + L, **kw)
            off = np.arange(0, (reads.shape[0] + 1) * L, L, dtype=np.int64)
            sm, ns, hits, nh, _ = O.seed_batch(idx, reads, off, smem_cap=512, hit_cap=1 << 15, threads=0)
            assert _gpu_dump(c, reads, off) == O.format_seed_dump(sm, ns, hits), L
    finally:
        c.close()


def test_index_files_loader_errors_and_equals_host_loader(g1, tmp_path):
    """meme_index_load_files streams the four files to the device through pinned pieces; the same failure modes as the
    reference's loader (message + error code instead of its exit, src/fastmap.cpp:430-433, 472-475), and the same index
    as meme_index_load_host on the same bytes."""
    import shutil
    c = hipapi.Context(0)
    try:
        with pytest.raises(hipapi.MemeError, match="cannot read"):
            c.load_index_files(str(tmp_path / "nothing_here"))
        broken = str(tmp_path / "broken")
        for ext in (".0123", ".pos_packed", ".suffixarray_uint64_L1_PARAMETERS", ".suffixarray_uint64_L2_PARAMETERS"):
            shutil.copy(g1 + ext, broken + ext)
        with open(broken + ".pos_packed", "ab") as fh:
            fh.write(b"\0" * 5)                                  # one suffix more than the text has bases
        with pytest.raises(hipapi.MemeError, match="disagree"):
            c.load_index_files(broken)
        with open(broken + ".pos_packed", "r+b") as fh:
            fh.truncate(os.path.getsize(g1 + ".pos_packed"))
        with open(broken + ".suffixarray_uint64_L2_PARAMETERS", "ab") as fh:
            fh.write(b"\0" * 24)                                 # no longer a power of two of records
        with pytest.raises(hipapi.MemeError, match="power-of-two"):
            c.load_index_files(broken)
        reads, off = read_fastq_codes(os.path.join(GOLDEN, "g1_reads_150.fq"))
        want = open(os.path.join(GOLDEN, "g1_seeds_150.txt")).read()
        c.load_index_host(np.fromfile(g1 + ".pos_packed", np.uint8), np.fromfile(g1 + ".0123", np.uint8),
                          np.fromfile(g1 + ".suffixarray_uint64_L1_PARAMETERS", np.uint8),
                          np.fromfile(g1 + ".suffixarray_uint64_L2_PARAMETERS", np.uint8))
        assert _gpu_dump(c, reads, off) == want
        c.load_index_files(g1)                                   # a ctx can be re-loaded
        assert _gpu_dump(c, reads, off) == want
    finally:
        c.close()


def test_reserved_buffers_change_nothing(g1):
    """meme_seed_reserve only allocates ahead of time (too little, enough, or far too much): same seeds through the pinned-result call."""
    for nres in (10, 140, 100_000):
        c = hipapi.Context(0)
        try:
            c.load_index_files(g1)
            c.seed_reserve(nres, nres * 150)
            reads, off = read_fastq_codes(os.path.join(GOLDEN, "g1_reads_150.fq"))
            smems, smem_off, hits, hit_off = c.seed_batch_host(reads, off)
            slots, counts, hl = hipapi.smems_to_slots(smems, smem_off, hits, hit_off)
            assert O.format_seed_dump(slots, counts, hl) == open(os.path.join(GOLDEN, "g1_seeds_150.txt")).read()
        finally:
            c.close()


def test_reseeding_on_the_plcp_table_equals_searching(tmp_path):
    """The re-seeding round of unique SMEMs runs off the search kernel (k_reseed: walks on the plcp table; k_reseed_emit / _search /
    _resume: batches of lane searches for what the table cannot answer) or, with seed_defer = 0, inside it as in round 3.  Same seeds
    either way and as the oracle: on a repeat-rich genome (regions that block at every step), on noisy 250-bp reads (many short
    SMEMs, regions that reach their SMEM's ends), on reads with N, with 16 SMEM slots per read (appended SMEMs overflow: the read goes
    to the next tier from the re-seeding kernels), and with the occurrence floors of other option sets."""
    from pymeme import workload
    g = synth.make_genome(8_000_000, seed=17, repeat_frac=0.10, repeat_len=350, n_families=6, divergence=0.02, n_dups=30, dup_len=3000, poly_runs=8)
    prefix = workload.build_index_on_disk(g, str(tmp_path), bits=0)
    idx = O.load_index_files(prefix)
    c = hipapi.Context(0)
    try:
        c.load_index_files(prefix)
        for L, kw in ((150, dict(n_frac=0.05)), (250, dict(sub_rate=0.05, indel_rate=0.0075)), (101, dict(exact_frac=0.3))):
            reads, _, _ = synth.make_reads(g, 4000, L, seed=500 + L, **kw)
            off = np.arange(0, (reads.shape[0] + 1) * L, L, dtype=np.int64)
            sm, ns, hits, nh, _ = O.seed_batch(idx, reads, off, smem_cap=1024, hit_cap=1 << 16, threads=0)
            want = O.format_seed_dump(sm, ns, hits)
            for defer, cap in ((1, 128), (0, 128), (1, 16)):
                c.set_tuning("seed_defer", defer)
                c.set_tuning("smem_cap", cap)
                assert _gpu_dump(c, reads, off) == want, (L, defer, cap)
                tm = c.timings()
                assert (tm.seed_reseed_ms > 0) == bool(defer)
        # other floors: min_seed_len / split factor / split width change which SMEMs are re-seeded and what a walk may emit
        reads, _, _ = synth.make_reads(g, 3000, 150, seed=77)
        off = np.arange(0, (reads.shape[0] + 1) * 150, 150, dtype=np.int64)
        c.set_tuning("smem_cap", 128)
        for msl, split_len, split_width in ((15, 20, 10), (25, 40, 3), (19, 28, 1)):
            opt = hipapi.default_seed_opt(rounds=3)
            opt.min_seed_len, opt.split_len, opt.split_width = msl, split_len, split_width
            p = O.default_seed_params(3)
            p.min_seed_len, p.split_len, p.split_width = msl, split_len, split_width
            sm, ns, hits, nh, _ = O.seed_batch(idx, reads, off, smem_cap=1024, hit_cap=1 << 16, threads=0, params=p)
            want = O.format_seed_dump(sm, ns, hits)
            for defer in (1, 0):
                c.set_tuning("seed_defer", defer)
                smems, smem_off, h, hit_off = c.seed_batch(reads, off, opt)
                slots, counts, hl = hipapi.smems_to_slots(smems, smem_off, h, hit_off)
                assert O.format_seed_dump(slots, counts, hl) == want, (msl, split_len, split_width, defer)
    finally:
        c.close()


def test_plcp_table_equals_definition(tmp_path):
    """meme_stage_build_plcp: plcp[u] = min(255, LCP of suffix u with the nearer of its two suffix-array neighbours), LCPs bounded by the
    text's end -- against a direct computation from the suffix array on a genome with long exact duplications (saturated entries)."""
    import torch
    g = synth.make_genome(400_000, seed=5, repeat_frac=0.05, n_dups=6, dup_len=600, poly_runs=4)
    fa = str(tmp_path / "p.fa")
    synth.write_fasta(fa, g, contigs=2)
    prefix = build_index(fa, bits=12)
    idx = O.load_index_files(prefix)
    text, sa = idx.text, idx.sa.astype(np.int64)
    n = text.shape[0]
    # LCP of adjacent suffixes by doubling comparison windows (capped at 256 > 255)
    lcp = np.zeros(n + 1, np.int64)                     # lcp[i] = LCP(slot i-1, slot i); lcp[0] = lcp[n] = 0
    a, b = sa[:-1], sa[1:]
    alive = np.ones(n - 1, bool)
    cur = np.zeros(n - 1, np.int64)
    pad = np.concatenate([text, np.full(300, 9, np.uint8)])
    pad2 = np.concatenate([text, np.full(300, 8, np.uint8)])  # different pads: running off the end never matches
    for k in range(256):
        ia = np.nonzero(alive)[0]
        if ia.size == 0:
            break
        same = pad[a[ia] + k] == pad2[b[ia] + k]
        cur[ia[same]] += 1
        alive[ia[~same]] = False
    lcp[1:n] = cur
    want = np.minimum(255, np.maximum(lcp[:-1], lcp[1:]))
    plcp_by_pos = np.zeros(n, np.int64)
    plcp_by_pos[sa] = want
    c = hipapi.Context(0)
    try:
        c.load_index_files(prefix)
        ia = c.describe_index()
        out = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        hipapi._check(hipapi.lib().meme_stage_build_plcp(hipapi.C.c_void_p(c.h), hipapi.C.c_void_p(ia.d_sa_ent), hipapi.C.c_int64(n), hipapi.C.c_void_p(ia.d_pac64),
                                                         hipapi.C.c_void_p(out.data_ptr())))
        c.sync()
        got = out[:n].cpu().numpy().astype(np.int64)
    finally:
        c.close()
    bad = np.nonzero(got != plcp_by_pos)[0]
    assert bad.size == 0, (int(bad[0]), int(got[bad[0]]), int(plcp_by_pos[bad[0]]))
    assert (got == 255).sum() > 500 and (got < 30).sum() > n // 2
