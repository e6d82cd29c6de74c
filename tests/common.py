"""Shared helpers for the test-suite (index building through our host tool, FASTQ parsing)."""
import os
import shutil
import subprocess
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
_TAB = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _TAB[_c] = _i
    _TAB[_c + 32] = _i

_CACHE = {}


def read_fastq_codes(path):
    """FASTQ -> (codes concatenated, offsets[n+1])"""
    seqs = []
    with open(path, "rb") as fh:
        lines = fh.read().split(b"\n")
    for i in range(1, len(lines), 4):
        if lines[i - 1].startswith(b"@"):
            seqs.append(_TAB[np.frombuffer(lines[i], dtype=np.uint8)])
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([s.shape[0] for s in seqs])
    return np.concatenate(seqs), off


def build_index(fasta, bits=12, threads=4):
    """Runs our meme-index on a copy of `fasta` in a temp dir; returns the index prefix (cached)."""
    key = (os.path.abspath(fasta), bits)
    if key in _CACHE:
        return _CACHE[key]
    d = tempfile.mkdtemp(prefix="memeidx_")
    dst = os.path.join(d, os.path.basename(fasta))
    shutil.copy(fasta, dst)
    subprocess.run([os.path.join(REPO, "bwa-meme_amd", "meme-index"), "build", dst, "-b", str(bits), "-t",
                    str(threads)], check=True, capture_output=True)
    _CACHE[key] = dst
    return dst


def chain_golden_workload():
    """The genome and reads tests/golden/chain_golden.npz was generated from (same seeds as tests/golden/make_chain_golden.py):
    returns (genome, list of reads as uint8 code arrays of their own lengths)."""
    from pymeme import synth
    g = synth.make_genome(300_000, seed=201, repeat_frac=0.15, repeat_len=250, n_families=5, divergence=0.02, n_dups=8, dup_len=1200, poly_runs=4)
    r1, _, _ = synth.make_reads(g, 2000, 150, seed=202, n_frac=0.03, exact_frac=0.2)
    r2, _, _ = synth.make_reads(g, 500, 250, seed=203, sub_rate=0.05, indel_rate=0.0075, n_frac=0.02)
    r3, _, _ = synth.make_reads(g, 300, 60, seed=204, sub_rate=0.02)
    reads, k = [], 0
    for rs in (r1, r2, r3):
        for r in rs:
            L = len(r) if rs is not r3 else 15 + k % 46
            reads.append(np.ascontiguousarray(r[:L], dtype=np.uint8))
            k += 1
    return g, reads


def gap_reads(g, n, seed):
    """Reads with an 80-95-base deletion or insertion next to a long exact stretch: the extension touches the band edge and is run
    again with the doubled band (src/bwamem.cpp:2985-3018)."""
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(n):
        p = int(rng.integers(1000, g.shape[0] - 2000))
        d = int(rng.integers(80, 96))
        if rng.random() < 0.5:
            r = np.concatenate([g[p:p + 110], g[p + 110 + d:p + 110 + d + 140]])
        else:
            r = np.concatenate([g[p:p + 90], rng.integers(0, 4, size=d).astype(np.uint8), g[p + 90:p + 90 + 160 - d]])
        r = r[:250].copy()
        if rng.random() < 0.5:
            r = (3 - r[::-1]).astype(np.uint8)
        rows.append(np.ascontiguousarray(r, dtype=np.uint8))
    return rows


def ext_golden_inputs():
    """Inputs of tests/golden/ext_golden.npz: the reads of the chain fixture + 300 gap reads, and their chains -- for the first 2 800
    reads the chains of tests/golden/chain_golden.npz (made by the compiled reference), for the gap reads the oracle's chains of the
    oracle's seeds (both pinned on the reference elsewhere).  Everything in the flat layout the batch calls use."""
    import oracle_py as O
    from pymeme import hipapi
    g, reads = chain_golden_workload()
    G = np.load(os.path.join(GOLDEN, "chain_golden.npz"))
    text = hipapi.fwd_rc_text(g)
    l_pac = int(G["l_pac"])
    n0 = len(reads)
    chains0 = np.zeros(G["chains"].shape[0], O.ORC_CHAIN_DTYPE)
    for k, f in enumerate(("pos", "rid", "n_seeds", "w", "kept", "first", "is_alt")):
        chains0[f] = G["chains"][:, k]
    per_chain = G["chains"][:, 2]
    cs = np.concatenate([[0], np.cumsum(per_chain)])
    seed_off0 = cs[G["chain_off"]]
    chains0["seed_beg"] = G["chains"][:, 7] - np.repeat(seed_off0[:-1], np.diff(G["chain_off"]))
    seeds0 = np.zeros(G["seeds"].shape[0], O.ORC_CSEED_DTYPE)
    seeds0["rbeg"], seeds0["qbeg"], seeds0["len"] = G["seeds"].T
    frac0 = G["frac_rep_bits"].view(np.float32)
    # gap reads: seeds and chains from the oracle (index = plain suffix array of the fixture's text)
    extra = gap_reads(g, 300, seed=205)
    from pymeme import hostapi
    _, sa = hostapi.build_sa(g)
    idx = O.Index(text, sa)
    eoff = np.zeros(len(extra) + 1, np.int64)
    eoff[1:] = np.cumsum([len(r) for r in extra])
    sm, nsm, hits, nh, _ = O.seed_batch(idx, np.concatenate(extra), eoff, smem_cap=512, hit_cap=1 << 14)
    copt = O.default_chain_opt(l_pac)
    alt = np.zeros(G["contig_off"].shape[0], np.uint8)
    ch_list, sd_list, frac1, coff1, soff1 = [], [], [], [0], [0]
    for r in range(len(extra)):
        rc, ch, sd, tree, frac = O.chain_read(sm[r, :nsm[r]], hits[r, :nh[r]], len(extra[r]), G["contig_off"], alt, copt)
        assert rc >= 0
        ch_list.append(ch); sd_list.append(sd); frac1.append(frac)
        coff1.append(coff1[-1] + rc); soff1.append(soff1[-1] + sd.shape[0])
    allreads = reads + extra
    read_off = np.zeros(len(allreads) + 1, np.int64)
    read_off[1:] = np.cumsum([len(r) for r in allreads])
    chains = np.concatenate([chains0] + ch_list)
    seeds = np.concatenate([seeds0] + sd_list)
    chain_off = np.concatenate([G["chain_off"], G["chain_off"][-1] + np.array(coff1[1:], np.int64)])
    seed_off = np.concatenate([seed_off0, seed_off0[-1] + np.array(soff1[1:], np.int64)])
    return {"genome": g, "reads_list": allreads, "reads": np.concatenate(allreads), "read_off": read_off, "chain_off": chain_off, "chains": chains,
            "seed_off": seed_off, "seeds": seeds, "frac_rep": np.concatenate([frac0, np.array(frac1, np.float32)]), "text": text, "l_pac": l_pac,
            "contig_off": G["contig_off"], "contig_len": G["contig_len"], "n_fixture_reads": n0}


def long_noisy_reads(g, n, seed, lo=800, hi=1300):
    """Reads of 800-1 300 bases from either strand with 4 % substitutions, short indels, an occasional N and -- what the seed filter's
    alignment is sensitive to -- stretches of 6-14 replaced bases (scored best as an insertion next to a deletion)."""
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        p = int(rng.integers(0, g.shape[0] - L - 200))
        src = g[p:p + L + 200]
        out, i = [], 0
        while len(out) < L and i < src.shape[0]:
            u = rng.random()
            if u < 0.004:                                          # replaced stretch
                k = int(rng.integers(6, 15))
                out += list(rng.integers(0, 4, size=k)); i += k
            elif u < 0.006:
                out += list(rng.integers(0, 4, size=int(rng.integers(1, 6))))
            elif u < 0.008:
                i += int(rng.integers(1, 6))
            else:
                c = int(src[i])
                if rng.random() < 0.04:
                    c = (c + int(rng.integers(1, 4))) & 3
                if rng.random() < 0.002:
                    c = 4
                out.append(c); i += 1
        r = np.array(out[:L], np.uint8)
        if rng.random() < 0.5:
            r = np.where(r < 4, 3 - r, 4)[::-1].astype(np.uint8)
        rows.append(np.ascontiguousarray(r))
    return rows


def flt_workload(min_chain_weight=0, n_long=120, n_short=400, lo=800, hi=1300):
    """Inputs for the seed filter (mem_flt_chained_seeds, src/bwamem.cpp:565-598): long noisy reads + 150 / 250-base reads of the chain
    fixture, on the chain fixture's genome; seeds and chains from the oracle (pinned on the reference elsewhere), chained with the given
    weight floor.  Without -W the filter runs for reads of ~790 bases and more -- which the function takes, but the learned-index path does
    not (LEARNED_MAX_READ_LEN 500, src/bwamem.cpp:1259-1262): lo / hi = 420 / 500 with -W 20 is the longest the aligner itself can meet.
    Layout of ext_golden_inputs()."""
    import oracle_py as O
    from pymeme import hipapi, hostapi
    g, fixture_reads = chain_golden_workload()
    G = np.load(os.path.join(GOLDEN, "chain_golden.npz"))
    text = hipapi.fwd_rc_text(g)
    l_pac = int(G["l_pac"])
    reads = long_noisy_reads(g, n_long, seed=311, lo=lo, hi=hi) + [fixture_reads[i] for i in list(range(0, n_short // 2)) + list(range(2000, 2000 + n_short // 2))]
    _, sa = hostapi.build_sa(g)
    idx = O.Index(text, sa)
    read_off = np.zeros(len(reads) + 1, np.int64)
    read_off[1:] = np.cumsum([len(r) for r in reads])
    sm, nsm, hits, nh, _ = O.seed_batch(idx, np.concatenate(reads), read_off, smem_cap=2048, hit_cap=1 << 15)
    copt = O.default_chain_opt(l_pac)
    copt.min_chain_weight = min_chain_weight
    alt = np.zeros(G["contig_off"].shape[0], np.uint8)
    ch_list, sd_list, frac, coff, soff = [], [], [], [0], [0]
    for r in range(len(reads)):
        rc, ch, sd, tree, fr = O.chain_read(sm[r, :nsm[r]], hits[r, :nh[r]], len(reads[r]), G["contig_off"], alt, copt)
        assert rc >= 0
        ch_list.append(ch); sd_list.append(sd); frac.append(fr)
        coff.append(coff[-1] + rc); soff.append(soff[-1] + sd.shape[0])
    return {"genome": g, "reads_list": reads, "reads": np.concatenate(reads), "read_off": read_off, "chain_off": np.array(coff, np.int64),
            "chains": np.concatenate(ch_list), "seed_off": np.array(soff, np.int64), "seeds": np.concatenate(sd_list), "frac_rep": np.array(frac, np.float32),
            "text": text, "l_pac": l_pac, "contig_off": G["contig_off"], "contig_len": G["contig_len"]}


def gcig_workload(n=3000, seed=77):
    """Inputs of the CIGAR-kernel tests (tests/golden/gcig_golden.npz): a 300 kbp genome, reads of 40-250 bases sampled with
    substitutions, indels (up to 30 bases) and an occasional N, from both strands; per read one or two global-alignment jobs the way
    mem_reg2aln (src/bwamem.cpp:2314-2380) would pose them -- a query span of the read against the text span it came from (fwd+rc
    coordinates; rev = both sequences reversed when on the reverse strand), band from 1 to 400.
    Returns (genome, reads list, jobs as hipapi.GJOB records, explicit (query, target) code arrays per job)."""
    from pymeme import hipapi, synth
    rng = np.random.default_rng(seed)
    g = synth.make_genome(300_000, seed=seed + 1, repeat_frac=0.05)
    text = hipapi.fwd_rc_text(g)
    l_pac = g.shape[0]
    reads, jobs, seqs = [], [], []
    for r in range(n):
        L = int(rng.integers(40, 251))
        strand = int(rng.integers(0, 2))
        p = int(rng.integers(0, 2 * l_pac - 2 * L - 64)) if False else int(rng.integers(0, l_pac - L - 64)) + strand * l_pac
        src = text[p:p + L + 40]
        out, i, tl = [], 0, 0
        rate_sub, rate_indel = rng.choice([0.0, 0.01, 0.05]), rng.choice([0.0, 0.003, 0.02])
        while len(out) < L:
            u = rng.random()
            if u < rate_indel / 2:                                  # insertion in the read
                out += list(rng.integers(0, 4, size=int(rng.integers(1, 31))))
            elif u < rate_indel:                                    # deletion from the read
                i += int(rng.integers(1, 31))
            else:
                c = int(src[min(i, src.shape[0] - 1)])
                if rng.random() < rate_sub:
                    c = (c + int(rng.integers(1, 4))) & 3
                if rng.random() < 0.002:
                    c = 4
                out.append(c); i += 1
        read = np.array(out[:L], np.uint8)
        tlen = int(min(max(i, 1), src.shape[0]))
        reads.append(read)
        for _ in range(1 + (r % 3 == 0)):
            qb = int(rng.integers(0, 8)) if rng.random() < 0.5 else 0
            qe = L - (int(rng.integers(0, 8)) if rng.random() < 0.5 else 0)
            rb = p + qb
            tl = max(1, min(tlen - qb + int(rng.integers(-3, 4)), 2 * l_pac - rb - 1, (l_pac - rb) if rb < l_pac else 10 ** 9))
            w = int(rng.choice([1, 3, 10, 30, 100, 200, 400]))
            w = max(w, abs(tl - (qe - qb)) + 3)                    # min_w of bwa_gen_cigar2 (src/bwa.cpp:314-315): the last cell is inside the band
            rev = 1 if rb >= l_pac else 0
            jobs.append((rb, r, qb, qe - qb, tl, w, rev))
            q, t = read[qb:qe], text[rb:rb + tl]
            seqs.append((q[::-1].copy(), t[::-1].copy()) if rev else (q.copy(), t.copy()))
    return g, reads, np.array(jobs, dtype=hipapi.GJOB), seqs


def kswv_workload(n=2000, seed=91, read_len=(60, 251), a=1, min_seed_len=19):
    """Mate-rescue jobs the way mem_matesw_batch_pre (src/bwamem_pair.cpp:1060-1223) poses them: a read (or its reverse complement) against a
    window of the genome of a few hundred bases that mostly contains its origin -- with substitutions, indels, an occasional N, windows
    that contain the read twice (tandem copies: the second-best score matters), windows that miss it, reads hanging over a window edge.
    xtra = KSW_XSUBO | KSW_XSTART | (len * a < 250 ? KSW_XBYTE : 0) | min_seed_len * a.  Returns (jobs KSWV_JOB_DTYPE, ref bytes, query bytes)."""
    from oracle_py import KSWV_JOB_DTYPE, KSW_XBYTE, KSW_XSUBO, KSW_XSTART
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=400_000, dtype=np.uint8)
    jobs = np.zeros(n, KSWV_JOB_DTYPE)
    refs, qers = [], []
    ro = qo = 0
    for k in range(n):
        L = int(rng.integers(read_len[0], read_len[1]))
        W = int(rng.integers(max(L // 2, min_seed_len), 900))
        p = int(rng.integers(0, g.shape[0] - 2000))
        win = g[p:p + W].copy()
        kind = k % 8
        off = int(rng.integers(0, max(1, W - L))) if W > L else 0
        if kind == 5:                                              # the read hangs over the window's edge
            off = max(0, W - L // 2)
        src = g[p + off:p + off + L + 40]
        out, i = [], 0
        rs, ri = (0.0, 0.0) if kind == 0 else (float(rng.choice([0.01, 0.03, 0.08])), float(rng.choice([0.0, 0.004, 0.02])))
        while len(out) < L:
            u = rng.random()
            if u < ri / 2:
                out += list(rng.integers(0, 4, size=int(rng.integers(1, 12))))
            elif u < ri:
                i += int(rng.integers(1, 12))
            else:
                c = int(src[min(i, src.shape[0] - 1)])
                if rng.random() < rs:
                    c = (c + int(rng.integers(1, 4))) & 3
                out.append(c); i += 1
        q = np.array(out[:L], np.uint8)
        if kind == 3:                                              # unrelated read: low scores, nothing above the threshold
            q = rng.integers(0, 4, size=L, dtype=np.uint8)
        if kind == 4 and W > 2 * L + 20:                           # a second, slightly worse copy of the origin further down the window
            cp = g[p + off:p + off + L].copy()
            m = rng.random(L) < 0.04
            cp[m] = (cp[m] + 1) & 3
            win[W - L - 5:W - 5] = cp
        if kind == 6:
            q[rng.integers(0, L, size=2)] = 4
            win[rng.integers(0, W, size=3)] = 4
        if kind == 7:                                              # low-complexity: many equal row maxima
            unit = rng.integers(0, 4, size=int(rng.integers(1, 4)), dtype=np.uint8)
            win[:] = np.resize(unit, W)
            q[:] = np.resize(unit, L)
            q[rng.integers(0, L, size=3)] = (q[0] + 1) & 3
        xtra = KSW_XSUBO | KSW_XSTART | (KSW_XBYTE if L * a < 250 else 0) | (min_seed_len * a)
        jobs[k] = (ro, qo, W, L, xtra, 0)
        refs.append(win); qers.append(q)
        ro += W; qo += L
    return jobs, np.concatenate(refs), np.concatenate(qers)


# (name, kswv_workload arguments, scoring parameters) of tests/golden/kswv_golden.npz
KSWV_GOLDEN_SETS = (
    ("default", dict(n=2500, seed=91), {}),
    ("long", dict(n=1500, seed=5, read_len=(200, 420)), {}),
    ("other", dict(n=1500, seed=6, read_len=(30, 300), a=2), dict(a=2, b=3, o_del=4, e_del=2, o_ins=5, e_ins=1)),
)


def kswv_edge_jobs():
    """Ten edge jobs of the mate-rescue kernel, every job with bytes of its own (as mem_matesw_batch_pre lays them out): the plain case, a
    one-base window, a window shorter than the read, a one-base query, no KSW_XSTART, no KSW_XSUBO, the int16 class, a caller-set KSW_XSTOP,
    a threshold the score does not reach, a 3 000-base window.  KSWV_EDGE_WANT: the compiled reference's kswr_t records for them."""
    from oracle_py import KSWV_JOB_DTYPE, KSW_XBYTE, KSW_XSTOP, KSW_XSUBO, KSW_XSTART
    rng = np.random.default_rng(3)
    g = rng.integers(0, 4, size=3000, dtype=np.uint8)
    q = g[1000:1150].copy()
    X = KSW_XSUBO | KSW_XSTART | KSW_XBYTE | 19
    spec = [(900, 400, 0, 150, X), (1000, 1, 0, 150, X), (1020, 60, 0, 150, X), (900, 400, 70, 1, X), (900, 400, 0, 150, KSW_XSUBO | KSW_XBYTE | 19),
            (900, 400, 0, 150, KSW_XSTART | KSW_XBYTE), (900, 400, 0, 150, KSW_XSTART | KSW_XSUBO | 19), (900, 400, 0, 150, KSW_XSTOP | KSW_XSUBO | 40),
            (900, 400, 0, 150, KSW_XSTART | KSW_XSUBO | KSW_XBYTE | 200), (0, 3000, 0, 150, X)]
    jobs = np.zeros(len(spec), KSWV_JOB_DTYPE)
    rb, qb = [], []
    ro = qo = 0
    for k, (idr, l1, idq, l2, x) in enumerate(spec):
        jobs[k] = (ro, qo, l1, l2, x, 0)
        rb.append(g[idr:idr + l1]); qb.append(q[idq:idq + l2])
        ro += l1; qo += l2
    return jobs, np.concatenate(rb), np.concatenate(qb)


KSWV_EDGE_WANT = [[150, 249, 149, -1, -1, 100, 0], [1, 0, 0, -1, -1, -1, -1], [60, 59, 79, -1, -1, 0, 20], [1, 4, 0, -1, -1, -1, -1], [150, 249, 149, -1, -1, -1, -1],
                  [150, 249, 149, -1, -1, 100, 0], [150, 249, 149, -1, -1, 100, 0], [40, 139, 39, -1, -1, -1, -1], [150, 249, 149, -1, -1, -1, -1],
                  [150, 1149, 149, -1, -1, 1000, 0]]


def gencig_workload(n=2500, seed=177):
    """Inputs of the bwa_gen_cigar2 tests (tests/golden/gencig_golden.npz): the reads and genome of gcig_workload's kind, but the jobs are CALLS of
    bwa_gen_cigar2 as mem_reg2aln makes them (src/bwamem.cpp:2340-2347) -- w_ is the band ARGUMENT (0 .. 400, the function derives the band),
    a share of the calls have equal lengths and w_ = 0 (the gap-free shortcut), spans start and end inside the read, both strands.
    Returns (genome, reads list, calls as hipapi.CJOB records)."""
    from pymeme import hipapi
    g, reads, jobs, _ = gcig_workload(n=n, seed=seed)
    rng = np.random.default_rng(seed + 5)
    l_pac = g.shape[0]
    calls = np.zeros(jobs.shape[0], dtype=hipapi.CJOB)
    for k, J in enumerate(jobs):
        qlen, tlen, rb = int(J["qlen"]), int(J["tlen"]), int(J["rb"])
        u = rng.random()
        if u < 0.35:                                   # equal lengths, w_ = 0: no DP
            tlen = qlen
            w_ = 0
        elif u < 0.5:                                  # equal lengths but a band: DP all the same
            tlen = qlen
            w_ = int(rng.choice([1, 5, 100]))
        else:
            w_ = int(rng.choice([0, 1, 3, 10, 30, 100, 200, 400]))
        tlen = max(1, min(tlen, (l_pac - rb) if rb < l_pac else 2 * l_pac - rb))
        calls[k] = (rb, int(J["read"]), int(J["qb"]), qlen, tlen, w_, 0)
    return g, reads, calls


def sam_workload(n=1500, seed=301, read_len=(30, 251)):
    """Inputs of the SAM-text tests (tests/golden/sam_golden.npz): records the way mem_reg2aln / mem_sam_pe leave them for mem_aln2sam -- mapped and
    unmapped ends, mates mapped / unmapped / on another contig / absent (single-end), both strands, CIGARs with clipping, insertions, deletions
    and their MD strings, secondary (0x100 and 0x10000) flags, supplementary records (which > 0: hard clipping unless is_alt or soft clipping is
    asked for), XA strings, with and without qualities.  Returns (recs SAM_REC_DTYPE, blob uint8, names list, reads list (codes), quals list
    (bytes or None), contig names)."""
    from oracle_py import SAM_REC_DTYPE
    rng = np.random.default_rng(seed)
    contigs = ["chr1", "chr2_random", "HLA-A*01:01", "c"]
    recs = np.zeros(n, SAM_REC_DTYPE)
    blob = bytearray()
    names, reads, quals = [], [], []

    def pad():
        while len(blob) % 4:
            blob.append(0)

    def cigar(L, clip_ok=True):
        ops, left = [], L
        if clip_ok and rng.random() < 0.4:
            c = int(rng.integers(1, max(2, L // 4))); ops.append((c, 3 if rng.random() < 0.8 else 4)); left -= c
        tail = None
        if clip_ok and rng.random() < 0.4 and left > 4:
            c = int(rng.integers(1, max(2, left // 4))); tail = (c, 3 if rng.random() < 0.8 else 4); left -= c
        while left > 0:
            m = int(rng.integers(1, left + 1)); ops.append((m, 0)); left -= m
            if left > 0 and rng.random() < 0.5:
                if rng.random() < 0.5:
                    i = int(rng.integers(1, min(left, 12) + 1)); ops.append((i, 1)); left -= i
                else:
                    ops.append((int(rng.integers(1, 40)), 2))
        if ops and ops[-1][1] == 2:
            ops.pop()
        if tail:
            ops.append(tail)
        return np.array([l << 4 | o for l, o in ops], np.uint32)

    def md_string():
        parts = []
        for _ in range(int(rng.integers(1, 6))):
            parts.append(str(int(rng.integers(0, 120))))
            parts.append("ACGT"[int(rng.integers(0, 4))] if rng.random() < 0.7 else "^" + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(1, 5)))))
        parts.append(str(int(rng.integers(0, 120))))
        return "".join(parts).encode()

    for k in range(n):
        L = int(rng.integers(*read_len))
        reads.append(rng.choice(np.array([0, 1, 2, 3, 4], np.uint8), size=L, p=[0.245, 0.245, 0.245, 0.245, 0.02]))
        quals.append(None if k % 11 == 0 else bytes(rng.integers(33, 74, size=L).astype(np.uint8)))
        names.append(("read_%d:%d/x" % (k, int(rng.integers(0, 10 ** 9)))).encode()[:int(rng.integers(1, 40))] or b"r")
        r = recs[k]
        r["read"] = k
        mapped = rng.random() < 0.9
        r["rid"] = int(rng.integers(0, len(contigs))) if mapped else -1
        r["pos"] = int(rng.integers(0, 3 * 10 ** 9)) if mapped else -1
        r["is_rev"] = int(rng.integers(0, 2)) if mapped else 0
        r["is_alt"] = int(rng.random() < 0.1)
        r["mapq"] = int(rng.integers(0, 61)) if mapped else 0
        r["flag"] = (0x40 if k & 1 else 0x80) | (2 if rng.random() < 0.6 else 0) | (0x100 if rng.random() < 0.05 else 0) | (0x10000 if rng.random() < 0.05 else 0) | \
                    (0x800 if rng.random() < 0.05 else 0) | (0x4 if not mapped else 0)
        r["which"] = int(rng.integers(1, 3)) if rng.random() < 0.15 else 0
        r["score"] = int(rng.integers(0, 251)) if mapped else -1 + int(rng.random() < 0.2)
        r["sub"] = int(rng.integers(-1, 200))
        r["NM"] = int(rng.integers(0, 40))
        if mapped:
            cg = cigar(L)
            pad(); r["cigar_off"] = len(blob); r["n_cigar"] = cg.shape[0]
            blob.extend(cg.tobytes()); blob.extend(md_string()); blob.append(0)
        u = rng.random()
        if u < 0.85:
            r["has_mate"] = 1
            m_mapped = rng.random() < 0.9
            r["m_rid"] = (int(r["rid"]) if (mapped and rng.random() < 0.8) else int(rng.integers(0, len(contigs)))) if m_mapped else -1
            r["m_pos"] = int(rng.integers(0, 3 * 10 ** 9)) if m_mapped else -1
            if m_mapped and mapped and rng.random() < 0.5:
                r["m_pos"] = int(r["pos"]) + int(rng.integers(-600, 600))
            r["m_is_rev"] = int(rng.integers(0, 2)) if m_mapped else 0
            r["m_is_alt"] = int(rng.random() < 0.1)
            if m_mapped and rng.random() < 0.95:
                mc = cigar(int(rng.integers(30, 251)))
                pad(); r["m_cigar_off"] = len(blob); r["m_n_cigar"] = mc.shape[0]
                blob.extend(mc.tobytes())
        r["xa_off"] = -1
        if mapped and rng.random() < 0.2:
            r["xa_off"] = len(blob)
            blob.extend(b"chr2_random,-%d,100M50S,%d;c,+17,150M,0;" % (int(rng.integers(1, 10 ** 8)), int(rng.integers(0, 30)))); blob.append(0)
    pad()
    return recs, np.frombuffer(bytes(blob), dtype=np.uint8).copy(), names, reads, quals, contigs


def matesw_pose_workload(seed=301, n_pairs=600, l_pac=240_000, n_contigs=3, read_len=(100, 251), pes=None):
    """Read pairs with alignment records the way worker_sam's first step meets them (after mem_sort_dedup_patch: best score first): most ends have a
    record at their origin, some have several (a near-best second one within pen_unpaired, low-scoring ones beyond it), some none; mates are
    consistent with the insert-size statistics, or on the wrong strand / too far / on another sequence / missing -- so that every branch of
    mem_matesw_batch_pre (src/bwamem_pair.cpp:1060-1223) is taken: orientations skipped because a mate record explains them, all four skipped,
    windows clipped at a sequence end or the strand boundary, windows that end up on another sequence (no job), windows shorter than min_seed_len.
    Returns dict(genome, text, contig_off, contig_len, reads, read_off, read_len, regs, reg_off, pes)."""
    from oracle_py import MATE_REG_DTYPE
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=l_pac, dtype=np.uint8)
    text = np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])
    cuts = np.sort(rng.choice(np.arange(2000, l_pac - 2000), size=n_contigs - 1, replace=False))
    contig_off = np.concatenate([[0], cuts]).astype(np.int64)
    contig_len = np.diff(np.concatenate([contig_off, [l_pac]])).astype(np.int32)
    if pes is None:
        pes = [(0, 0, 1), (120, 680, 0), (0, 0, 1), (0, 0, 1)]                 # the usual paired-end library: only FR passes
    reads, lens, regs, reg_off = [], [], [], [0]
    for p in range(n_pairs):
        L1, L2 = int(rng.integers(*read_len)), int(rng.integers(*read_len))
        ins = int(rng.integers(150, 650))
        kind = p % 12
        pos = int(rng.integers(0, l_pac - 1500))
        if kind == 9:
            pos = int(rng.choice([0, 30, l_pac - 400, int(contig_off[1]) - 200, int(contig_off[1]) + 5]))       # windows clipped at sequence ends
            pos = max(0, min(pos, l_pac - 300))
        strand = int(rng.integers(0, 2))
        ends = []
        for e, Lr in ((0, L1), (1, L2)):
            fpos = pos if e == 0 else min(l_pac - Lr - 1, pos + ins)
            rd = g[fpos:fpos + Lr].copy()
            m = rng.random(Lr) < 0.02
            rd[m] = (rd[m] + 1) & 3
            if rng.random() < 0.05:
                rd[rng.integers(0, Lr)] = 4
            rev = (e == 1) ^ (strand == 1)
            if rev:
                rd = np.where(rd < 4, 3 - rd, 4)[::-1].astype(np.uint8)
            rb = (2 * l_pac - (fpos + Lr)) if rev else fpos
            ends.append((rd, rb, Lr, fpos))
        for e, (rd, rb, Lr, fpos) in enumerate(ends):
            reads.append(rd); lens.append(Lr)
            recs = []
            rid = int(np.searchsorted(contig_off, fpos, side="right") - 1)
            if not (kind == 1 and e == 1) and not (kind == 2):                   # kind 1: the mate has no record; kind 2: neither end has
                recs.append((rb, rid, int(Lr - rng.integers(0, 12))))
            if kind == 3 and e == 1:                                             # the mate lies on the wrong strand / far away / on another sequence
                far = int(rng.integers(0, 2 * l_pac - 300))
                frid = int(np.searchsorted(contig_off, far if far < l_pac else 2 * l_pac - 1 - far, side="right") - 1)
                recs = [(far, frid, int(Lr - 5))]
            if kind in (4, 5, 6):                                                # several records: near-best ones are looked at too, the rest are not
                for _ in range(int(rng.integers(1, 5))):
                    far = int(rng.integers(0, 2 * l_pac - 300))
                    frid = int(np.searchsorted(contig_off, far if far < l_pac else 2 * l_pac - 1 - far, side="right") - 1)
                    recs.append((far, frid, int(Lr - rng.integers(0, 40))))
            if kind == 7 and recs:
                recs[0] = (recs[0][0], (recs[0][1] + 1) % n_contigs, recs[0][2])     # a record whose rid is not the window's sequence: no job
            recs.sort(key=lambda x: -x[2])
            regs += recs
            reg_off.append(len(regs))
    read_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    R = np.zeros(len(regs), MATE_REG_DTYPE)
    for k, (rb, rid, sc) in enumerate(regs):
        R[k] = (rb, rid, sc)
    return dict(genome=g, text=text, l_pac=l_pac, contig_off=contig_off, contig_len=contig_len, reads=np.concatenate(reads).astype(np.uint8), read_off=read_off,
                read_len=np.array(lens, np.int32), regs=R, reg_off=np.array(reg_off, np.int64), pes=np.array(pes, np.int32))
