"""End-to-end acceptance test of the drop-in (the reference's own acceptance criterion, README.md:81-92):
the reference aligner with seeding and extension interposed by the HIP backend (oracle/_ref/bwa-meme_dropin =
reference main + libbwa_pic.so + bwa-meme_amd/binding/meme_dropin*.cpp over include/meme_hip.h) must write the same SAM
as the unmodified reference binary (`mem -7`), apart from the @PG line that embeds the command line."""
import os
import subprocess

import numpy as np
import pytest

import ref_py as R
from common import build_index
from pymeme import synth

pytestmark = pytest.mark.gpu

REF = R.REF_DIR
# the mate-rescue stage is opt-in (it does not pay at the job counts real chunks pose) and then only engages on chunks that pose tens of
# thousands of Smith-Waterman jobs: here every paired-end run uses it
os.environ.setdefault("MEME_DROPIN_MATESW", "1")
os.environ.setdefault("MEME_DROPIN_MATESW_MIN", "0")
# every record the device formats (SAM text) is also formatted by the reference's mem_aln2sam and compared inside the aligner
os.environ.setdefault("MEME_DROPIN_SAM_CHECK", "1")
# round 6: mate rescue is posed on the device; in the tests the reference's own posing function runs over the same records as well and every job index / result is compared in the aligner
os.environ.setdefault("MEME_DROPIN_MATE_CHECK", "1")


def _sam(exe, prefix, fqs, env=None, threads=4, chunk=100000000, opts=(), stderr=None, timeout=900):
    cmd = [os.path.join(REF, exe), "mem", "-7", "-Y", "-K", str(chunk), "-t", str(threads)] + list(opts) + [prefix] + fqs
    r = subprocess.run(cmd, capture_output=True, env=env, timeout=timeout)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    if stderr is not None:
        stderr.append(r.stderr.decode())
    return [l for l in r.stdout.decode().split("\n") if not l.startswith("@PG")]


@pytest.mark.skipif(not (R.have("bwa-meme_dropin") and R.have("bwa-meme_mode3") and R.cpu_can_run()),
                    reason="compiled reference (oracle/_ref) not available on this box")
@pytest.mark.parametrize("paired", [False, True])
def test_sam_identical_to_reference(tmp_path, paired):
    g = synth.make_genome(400_000, seed=41, repeat_frac=0.08, n_families=6, n_dups=6, dup_len=1500)
    fa = str(tmp_path / "e2e.fa")
    synth.write_fasta(fa, g, contigs=3)
    prefix = build_index(fa, bits=14)
    n = 6000
    r1, pos, strand = synth.make_reads(g, n, 150, seed=42, n_frac=0.03, exact_frac=0.2)
    fqs = [str(tmp_path / "r1.fq")]
    synth.write_fastq(fqs[0], r1, prefix="p")
    if paired:
        # mates: reverse-complement reads ~350 bp downstream on the same fragment
        rng = np.random.default_rng(43)
        ins = rng.integers(300, 500, size=n)
        p2 = np.clip(pos + ins - 150, 0, g.shape[0] - 160)
        idx = p2[:, None] + np.arange(150)[None, :]
        r2 = 3 - g[idx][:, ::-1]
        sub = rng.random(r2.shape) < 0.01
        r2 = np.where(sub, (r2 + 1) & 3, r2).astype(np.uint8)
        # keep name pairing: same read names in both files
        fqs.append(str(tmp_path / "r2.fq"))
        synth.write_fastq(fqs[1], r2, prefix="p")
    want = _sam("bwa-meme_mode3", prefix, fqs)
    # chunk-level seeding + combined extension calls: the SAM must not depend on how many worker threads feed the
    # combiner, nor on the -K chunk size (several chunks per run, the last one ragged)
    # Chaining + seed filter + extension run on the device for the whole chunk (default); the cross-check is the reference's own
    # per-batch function over the combiner (MEME_DROPIN_EXT=0).  MEME_DROPIN_VIRTUAL=3 runs the
    # multi-GPU arrangement (three device slots: reads of a chunk split three ways, index replicas, one extension call per slot)
    # on however many GPUs the box has.  Chaining runs on the device (mem_chain_Learned + mem_chain_flt) with
    # MEME_DROPIN_CHAIN_CHECK set: every read is also chained by the reference's host functions and any difference in any
    # chain or seed is fatal; MEME_DROPIN_CHAIN=0 keeps chaining on the host, MEME_DROPIN_IO=1 lets the binding parse the two
    # FASTQ files on two threads.
    small = {}
    # Round 3: the DEFAULT is chaining + extension on the device (meme_extend_last_batch_host: the host only receives alignment records).
    # Round 4: the device stages of chunk k+1 run beside the SAM phase of chunk k (prefetch; also switched off), and a backend that refuses
    # a batch for want of memory (MEME_DROPIN_MAX_BATCH: max_batch of the ctxs) is fed in pieces -- extension stage and CIGAR stage alike.
    for threads, chunk, extra in ((4, 100000000, {}), (16, 400000, {}), (8, 400000, {"MEME_DROPIN_VIRTUAL": "3"}),
                                  (8, 400000, {"MEME_DROPIN_PREFETCH": "0"}), (8, 400000, {"MEME_DROPIN_SAM": "0"}), (8, 100000000, {"MEME_DROPIN_MAX_BATCH": "1500"}),
                                  # round 6 (advisor): a SAM-text call the device refuses is repeated in halves (3 000-slot pieces of a 6 000 / 12 000-read chunk) and what
                                  # it refuses even in pieces of 1 024 is formatted by the reference's own mem_aln2sam from the noted descriptors -- same SAM either way
                                  (8, 100000000, {"MEME_DROPIN_SAM_MAX_BATCH": "3000"}), (8, 400000, {"MEME_DROPIN_SAM_MAX_BATCH": "500", "MEME_DROPIN_VIRTUAL": "2"}),
                                  (8, 400000, {"MEME_DROPIN_MAX_BATCH": "700", "MEME_DROPIN_VIRTUAL": "2"}),
                                  (16, 400000, {"MEME_DROPIN_EXT": "0"}),
                                  (4, 100000000, {"MEME_DROPIN_EXT": "0", "MEME_DROPIN_CHAIN": "0", "MEME_DROPIN_IO": "1"})):
        env = dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_CHAIN_CHECK="1", **extra)
        got = _sam("bwa-meme_dropin", prefix, fqs, env=env, threads=threads, chunk=chunk)
        if chunk == 100000000: ref = want
        else:
            if "ref" not in small: small["ref"] = _sam("bwa-meme_mode3", prefix, fqs, threads=threads, chunk=chunk)
            ref = small["ref"]
        assert len(got) == len(ref) and len(ref) > n
        diff = [(a, b) for a, b in zip(got, ref) if a != b]
        assert not diff, "threads=%d chunk=%d %r, first differing SAM line:\n%s\n%s" % ((threads, chunk, extra) + diff[0])
    # a value of MEME_DROPIN_EXT that no longer exists (the host-side extension stage of rounds 2-3) stops the run before the index loads
    r = subprocess.run([os.path.join(REF, "bwa-meme_dropin"), "mem", "-7", "-t", "2", prefix] + fqs, capture_output=True,
                       env=dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_EXT="host"), timeout=300)
    assert r.returncode == 1 and b"the values are device and 0" in r.stderr, r.stderr.decode()[-500:]


@pytest.mark.skipif(not (R.have("bwa-meme_dropin") and R.have("bwa-meme_mode3") and R.cpu_can_run()),
                    reason="compiled reference (oracle/_ref) not available on this box")
def test_sam_identical_with_given_insert_size(tmp_path):
    """`mem -I mean,sd`: the insert-size distribution is given, mem_pestat is not called (reference src/fastmap.cpp:1430-1450, src/bwamem.cpp:1950-1959) -- the binding's
    record digest is then built by whichever pre-pass of the SAM phase arrives first (under a lock since round 6), and mate rescue is posed with the given bounds."""
    g = synth.make_genome(400_000, seed=41, repeat_frac=0.08, n_families=6, n_dups=6, dup_len=1500)
    fa = str(tmp_path / "ins.fa")
    synth.write_fasta(fa, g, contigs=3)
    prefix = build_index(fa, bits=14)
    n = 5000
    a, b = _pe_reads(g, n, 150, seed=61, sub=0.02, indel=0.002)
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    synth.write_fastq(f1, a, prefix="i")
    synth.write_fastq(f2, b, prefix="i")
    opts = ["-I", "400,60"]
    want = _sam("bwa-meme_mode3", prefix, [f1, f2], threads=8, chunk=400000, opts=opts, timeout=240)
    got = _sam("bwa-meme_dropin", prefix, [f1, f2], env=dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_CHAIN_CHECK="1"), threads=8, chunk=400000, opts=opts, timeout=240)
    assert len(got) == len(want) and len(want) > 2 * n
    diff = [(x, y) for x, y in zip(got, want) if x != y]
    assert not diff, "first differing SAM line:\n%s\n%s" % diff[0]


@pytest.mark.skipif(not (R.have("bwa-meme_dropin") and R.have("bwa-meme_mode3") and R.cpu_can_run()),
                    reason="compiled reference (oracle/_ref) not available on this box")
def test_sam_identical_ecoli_sized_100k_reads(tmp_path):
    """BASELINE.json configs[0]: an E. coli K-12 sized reference (4.64 Mbp, one contig) and 100 k synthetic 150-bp
    single-end reads through `mem -7`: SAM diff == empty."""
    g = synth.make_genome(4_641_652, seed=7, repeat_frac=0.03, n_families=8, n_dups=7, dup_len=1200)
    fa = str(tmp_path / "ecoli_sized.fa")
    synth.write_fasta(fa, g, contigs=1)
    prefix = build_index(fa, bits=18, threads=min(32, os.cpu_count() or 4))
    n = 100_000
    r1, _, _ = synth.make_reads(g, n, 150, seed=8, n_frac=0.01, exact_frac=0.2)
    fq = str(tmp_path / "r.fq")
    synth.write_fastq(fq, r1, prefix="e")
    threads = min(32, os.cpu_count() or 4)
    want = _sam("bwa-meme_mode3", prefix, [fq], threads=threads)
    got = _sam("bwa-meme_dropin", prefix, [fq], env=dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_CHAIN_CHECK="1"), threads=threads)
    assert len(got) == len(want) and len(want) > n
    diff = [(a, b) for a, b in zip(got, want) if a != b]
    assert not diff, "first differing SAM line:\n%s\n%s" % diff[0]


@pytest.mark.skipif(not (R.have("bwa-meme_dropin") and R.have("bwa-meme_mode3") and R.cpu_can_run()),
                    reason="compiled reference (oracle/_ref) not available on this box")
@pytest.mark.parametrize("ext_on_device", ["device", "0"])
def test_sam_identical_with_lower_case_letters(tmp_path, ext_on_device):
    """FASTQ letters in lower case (soft-masked input): whole reads, runs, single letters, `n` among them -- nst_nt4_table maps both cases (reference
    src/bntseq.cpp:63-80, applied in src/bwamem.cpp:1277-1279).  The binding converts 64 letters at a time where they are all A C G T N of either case and by the
    table otherwise; the device converts the raw letters itself.
    NOT in this test: IUPAC ambiguity letters (R, Y, ...).  The reference decides whether a read has an N by looking for the characters 'N' and 'n'
    (src/bwamem.cpp:1255-1258) while every other letter becomes base 4 as well: its N-free search path then never ends (observed with `mem -7` on one read
    with an R: two worker threads spin in worker_bwt for good).  There is no reference output to be identical to; DESIGN 7."""
    g = synth.make_genome(300_000, seed=51, repeat_frac=0.05, n_families=4, n_dups=4, dup_len=900)
    fa = str(tmp_path / "lc.fa")
    synth.write_fasta(fa, g, contigs=2)
    prefix = build_index(fa, bits=14)
    n = 3000
    r1, _, _ = synth.make_reads(g, n, 151, seed=52, n_frac=0.02, exact_frac=0.3)
    fq = str(tmp_path / "r.fq")
    synth.write_fastq(fq, r1, prefix="l")
    rng = np.random.default_rng(53)
    lines = open(fq, "rb").read().split(b"\n")
    for i in range(n):
        s = bytearray(lines[4 * i + 1])
        u = i % 6
        if u == 0: s = bytearray(bytes(s).lower())                                     # a whole read in lower case
        elif u == 1:                                                                    # a lower-case (soft-masked) run
            p = int(rng.integers(0, 120)); s[p:p + 30] = bytes(s[p:p + 30]).lower()
        elif u == 2:                                                                    # single lower-case letters, an n among them
            for p in rng.integers(0, len(s), size=3): s[int(p)] = bytes(s[int(p):int(p) + 1]).lower()[0]
            s[int(rng.integers(0, len(s)))] = ord("n")
        elif u == 3:                                                                    # one in the last (scalar) stretch and one in the first 64-letter block
            p = len(s) - 1 - int(rng.integers(0, 20)); s[p] = bytes(s[p:p + 1]).lower()[0]
            s[int(rng.integers(0, 64))] = ord("n")
        lines[4 * i + 1] = bytes(s)
    open(fq, "wb").write(b"\n".join(lines))
    assert not any(c in b"RYKMSWBDHVryk.-*xu" for l in lines[1::4] for c in l)          # (see above: the reference does not return from such a read)
    want = _sam("bwa-meme_mode3", prefix, [fq], timeout=240)
    got = _sam("bwa-meme_dropin", prefix, [fq], env=dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_CHAIN_CHECK="1", MEME_DROPIN_EXT=ext_on_device), timeout=240)
    assert len(got) == len(want) and len(want) > n
    diff = [(a, b) for a, b in zip(got, want) if a != b]
    assert not diff, "first differing SAM line:\n%s\n%s" % diff[0]


@pytest.mark.skipif(not (R.have("bwa-meme_dropin") and R.have("bwa-meme_mode3") and R.cpu_can_run()),
                    reason="compiled reference (oracle/_ref) not available on this box")
def test_sam_identical_with_long_gaps_and_short_reads(tmp_path):
    """Reads that need the second band width (an 80-95-base deletion or insertion next to the seed: max_off >= 3w/4, so the job
    is run again with w = 200, src/bwamem.cpp:2985-3018), 250-bp reads with 5 % substitutions (BASELINE configs[5]) and reads
    too short to seed, mixed in one run; the chunk-wide extension stage must report that it took the retry path."""
    import re
    g = synth.make_genome(600_000, seed=51, repeat_frac=0.05, n_families=4, n_dups=4, dup_len=1000)
    fa = str(tmp_path / "gaps.fa")
    synth.write_fasta(fa, g, contigs=2)
    prefix = build_index(fa, bits=14)
    rng = np.random.default_rng(52)
    rows = []
    for _ in range(1500):
        p = int(rng.integers(1000, g.shape[0] - 2000))
        d = int(rng.integers(80, 96))
        if rng.random() < 0.5:                                   # deletion in the read: two stretches of the genome, d bases apart
            r = np.concatenate([g[p:p + 110], g[p + 110 + d:p + 110 + d + 140]])
        else:                                                    # insertion of d random bases
            r = np.concatenate([g[p:p + 90], rng.integers(0, 4, size=d).astype(np.uint8), g[p + 90:p + 90 + 160 - d]])
        r = r[:250].copy()
        if rng.random() < 0.5:
            r = (3 - r[::-1]).astype(np.uint8)
        rows.append(r)
    noisy, _, _ = synth.make_reads(g, 1500, 250, seed=53, sub_rate=0.05, indel_rate=0.0075, n_frac=0.02)
    short, _, _ = synth.make_reads(g, 300, 250, seed=54)
    # pathological reads: homopolymers and a dinucleotide run (thousands of SMEMs and hits where the genome has such runs: overflow
    # tiers of the search kernel, more SMEMs than the chain kernel takes), all N, N every 20 bases, one base
    weird = [np.zeros(150, np.uint8), np.full(150, 3, np.uint8), np.tile(np.array([0, 1], np.uint8), 75), np.full(150, 4, np.uint8),
             np.where(np.arange(150) % 20 == 7, 4, g[5000:5150]).astype(np.uint8), g[7000:7001].copy()]
    rows += weird
    fq = str(tmp_path / "gaps.fq")
    with open(fq, "w") as fh:
        k = 0
        for r in rows + list(noisy):
            fh.write("@g%d\n%s\n+\n%s\n" % (k, "".join("ACGTN"[c] for c in r), "I" * len(r)))
            k += 1
        for r in short:                                          # 12 - 30 bases: below or just above min_seed_len
            L = 12 + k % 19
            fh.write("@g%d\n%s\n+\n%s\n" % (k, "".join("ACGTN"[c] for c in r[:L]), "I" * L))
            k += 1
    want = _sam("bwa-meme_mode3", prefix, [fq], threads=8)
    cmd = [os.path.join(REF, "bwa-meme_dropin"), "mem", "-7", "-Y", "-K", "100000000", "-t", "8", prefix, fq]
    r = subprocess.run(cmd, capture_output=True, env=dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_VERBOSE="1", MEME_DROPIN_CHAIN_CHECK="1"), timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = [l for l in r.stdout.decode().split("\n") if not l.startswith("@PG")]
    assert len(got) == len(want) and len(want) > 3306
    diff = [(a, b) for a, b in zip(got, want) if a != b]
    assert not diff, "first differing SAM line:\n%s\n%s" % diff[0]
    m = re.search(r"\((\d+) of them again with the doubled band\)", r.stderr.decode())
    assert m and int(m.group(1)) > 100, r.stderr.decode()[-1500:]


@pytest.mark.skipif(not (R.have("bwa-meme_dropin") and R.have("bwa-meme_mode3") and R.cpu_can_run()),
                    reason="compiled reference (oracle/_ref) not available on this box")
def test_sam_identical_where_the_seed_filter_runs(tmp_path):
    """mem_flt_chained_seeds (src/bwamem.cpp:565-598) is a no-op unless 1.1 x the -W chain weight floor <= 0.05 x read length -- never
    without -W for reads the aligner takes (its reader keeps 301 bases of a longer read).  Here it is not: 150 / 250-base reads under
    -W 5 and -W 10 and noisy long reads under -W 13 (also with 12-base seeds and other penalties).  In the aligner the function reads its
    windows from the bit-reversed rc_pac (see flt_window_base in csrc/meme_kswv.hip), so the alignments score what unrelated sequences
    score: above the bar of -W 5 (the seeds stay, with that score: the order they are extended in changes), below the others (seeds short
    enough to be aligned leave their chains).  On the device by default -- the run reports the filter's alignments -- and through the
    reference's own function (MEME_DROPIN_EXT=0) as the cross-check."""
    import re
    from common import long_noisy_reads
    g = synth.make_genome(600_000, seed=61, repeat_frac=0.08, n_families=4, n_dups=6, dup_len=1500)
    fa = str(tmp_path / "flt.fa")
    synth.write_fasta(fa, g, contigs=3)
    prefix = build_index(fa, bits=14)
    long_fq, short_fq = str(tmp_path / "long.fq"), str(tmp_path / "short.fq")
    with open(long_fq, "w") as fh:
        for k, r in enumerate(long_noisy_reads(g, 1500, seed=62, lo=301, hi=420)):
            fh.write("@L%d\n%s\n+\n%s\n" % (k, "".join("ACGTN"[c] for c in r), "I" * len(r)))
    r150, _, _ = synth.make_reads(g, 2500, 150, seed=63, n_frac=0.02)
    r250, _, _ = synth.make_reads(g, 1500, 250, seed=64, sub_rate=0.05, indel_rate=0.0075, n_frac=0.02)
    with open(short_fq, "w") as fh:
        for k, r in enumerate(list(r150) + list(r250)):
            fh.write("@s%d\n%s\n+\n%s\n" % (k, "".join("ACGTN"[c] for c in r), "I" * len(r)))
    for fq, opts, drops in ((short_fq, ("-W", "5"), None), (short_fq, ("-W", "10"), True), (long_fq, ("-W", "13", "-k", "12"), True),
                            (long_fq, ("-W", "13", "-k", "12", "-A", "2", "-B", "7"), True)):
        want = _sam("bwa-meme_mode3", prefix, [fq], threads=8, opts=opts)
        for extra in ({}, {"MEME_DROPIN_EXT": "0"}):
            err = []
            got = _sam("bwa-meme_dropin", prefix, [fq], env=dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_VERBOSE="1", **extra), threads=8, opts=opts, stderr=err)
            assert len(got) == len(want) and len(want) > 1500
            diff = [(a, b) for a, b in zip(got, want) if a != b]
            assert not diff, "%r %r: first differing SAM line:\n%s\n%s" % ((opts, extra) + diff[0])
            if not extra:
                m = re.search(r"seed filter \(mem_flt_chained_seeds\) on the device: (\d+) alignments, (\d+) chained seeds removed", err[0])
                assert m and int(m.group(1)) > 1000 and (drops is None or (int(m.group(2)) > 0) == drops), (opts, err[0][-1500:])


@pytest.mark.skipif(not (R.have("bwa-meme_dropin") and R.have("bwa-meme_mode3") and R.cpu_can_run()),
                    reason="compiled reference (oracle/_ref) not available on this box")
def test_sam_identical_with_alt_contigs(tmp_path):
    """A reference with ALT contigs (<prefix>.alt, src/bntseq.cpp:208-239): the third contig is a diverged copy of a stretch of the
    first, so reads from that stretch chain equally well on both and the chain filter's ALT rule decides (src/bwamem.cpp:666);
    chained on the device and, with MEME_DROPIN_CHAIN_CHECK, on the host."""
    g = synth.make_genome(300_000, seed=61, repeat_frac=0.02)
    third = 100_000
    alt = g[20_000:20_000 + third].copy()
    rng = np.random.default_rng(62)
    flip = rng.random(third) < 0.004
    alt[flip] = (alt[flip] + 1) & 3
    g = np.concatenate([g[:2 * third], alt])
    fa = str(tmp_path / "alt.fa")
    synth.write_fasta(fa, g, contigs=3)
    prefix = build_index(fa, bits=14)
    with open(prefix + ".alt", "w") as fh:
        fh.write("chrS3\n")
    r1, _, _ = synth.make_reads(g[:2 * third], 4000, 150, seed=63, n_frac=0.01)
    fq = str(tmp_path / "alt.fq")
    synth.write_fastq(fq, r1, prefix="a")
    want = _sam("bwa-meme_mode3", prefix, [fq], threads=8)
    got = _sam("bwa-meme_dropin", prefix, [fq], env=dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_CHAIN_CHECK="1"), threads=8)
    assert len(got) == len(want) and len(want) > 4000
    diff = [(a, b) for a, b in zip(got, want) if a != b]
    assert not diff, "first differing SAM line:\n%s\n%s" % diff[0]
    assert sum(1 for l in want if "\tchrS3\t" in l or "chrS3," in l) > 100           # the ALT contig does take part


@pytest.mark.skipif(not (R.have("bwa-meme_dropin") and R.have("bwa-meme_mode3") and R.cpu_can_run()),
                    reason="compiled reference (oracle/_ref) not available on this box")
@pytest.mark.parametrize("short_file", [1, 2])
def test_sam_identical_when_one_mate_file_is_shorter(tmp_path, short_file):
    """The binding's FASTQ parser (one thread per mate file) must stop where the reference's does when the files differ in length
    (src/bwa.cpp:190-193, 223-226), across several chunks."""
    g = synth.make_genome(200_000, seed=71)
    fa = str(tmp_path / "u.fa")
    synth.write_fasta(fa, g, contigs=2)
    prefix = build_index(fa, bits=13)
    n = 3000
    r1, pos, _ = synth.make_reads(g, n, 100, seed=72, n_frac=0.0)
    r2, _, _ = synth.make_reads(g, n, 100, seed=73, n_frac=0.0)
    f1, f2 = str(tmp_path / "u1.fq"), str(tmp_path / "u2.fq")
    synth.write_fastq(f1, r1[:n - 37] if short_file == 1 else r1, prefix="u")
    synth.write_fastq(f2, r2[:n - 37] if short_file == 2 else r2, prefix="u")
    want = _sam("bwa-meme_mode3", prefix, [f1, f2], threads=4, chunk=100000)
    got = _sam("bwa-meme_dropin", prefix, [f1, f2], env=dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_IO="1"), threads=4, chunk=100000)
    assert len(got) == len(want) and len(want) > 2 * (n - 40)
    diff = [(a, b) for a, b in zip(got, want) if a != b]
    assert not diff, "first differing SAM line:\n%s\n%s" % diff[0]


def _pe_reads(g, n, length, seed, sub, indel, n_frac=0.01):
    """n read pairs of one length: mate 1 forward, mate 2 the reverse complement 300-500 bases downstream, both with errors."""
    rng = np.random.default_rng(seed)
    pos = rng.integers(0, g.shape[0] - 700 - length, size=n)
    ins = rng.integers(300, 500, size=n) + (length - 150)

    def noisy(x):
        out = []
        for row in x:
            r = []
            i = 0
            while len(r) < length and i < row.shape[0]:
                u = rng.random()
                if u < indel / 2: r.append(int(rng.integers(0, 4)))                 # insertion
                elif u < indel: i += 1                                               # deletion
                else:
                    c = int(row[i]); i += 1
                    if rng.random() < sub: c = (c + int(rng.integers(1, 4))) & 3
                    if rng.random() < n_frac / length * 3: c = 4
                    r.append(c)
            while len(r) < length: r.append(int(rng.integers(0, 4)))
            out.append(r)
        return np.array(out, np.uint8)
    ar = np.arange(length + 40)
    a = noisy(g[pos[:, None] + ar[None, :]])
    b = noisy((3 - g[(pos + ins - length - 40)[:, None] + ar[None, :]][:, ::-1]).astype(np.uint8))
    return a, b


@pytest.mark.skipif(not (R.have("bwa-meme_dropin") and R.have("bwa-meme_mode3") and R.cpu_can_run()),
                    reason="compiled reference (oracle/_ref) not available on this box")
def test_sam_identical_mixed_250bp_high_error_paired(tmp_path):
    """BASELINE configs[4]'s read class as a PAIRED run: half 150-bp pairs with 1 % substitutions, half 250-bp pairs with 5 % substitutions
    and 0.75 % indels, mixed in one pair of files (several -K chunks): long banded-DP jobs, SMEM divergence, band retries, mate rescue, and
    the CIGAR stage (most alignments need the global alignment with gaps)."""
    import re
    g = synth.make_genome(1_500_000, seed=81, repeat_frac=0.06, n_families=6, n_dups=6, dup_len=1500)
    fa = str(tmp_path / "mix.fa")
    synth.write_fasta(fa, g, contigs=4)
    prefix = build_index(fa, bits=16)
    a150, b150 = _pe_reads(g, 3000, 150, 82, 0.01, 0.0015)
    a250, b250 = _pe_reads(g, 3000, 250, 83, 0.05, 0.0075)
    order = np.random.default_rng(84).permutation(6000)
    f1, f2 = str(tmp_path / "m1.fq"), str(tmp_path / "m2.fq")
    with open(f1, "w") as h1, open(f2, "w") as h2:
        for k in order:
            ra, rb = (a150[k], b150[k]) if k < 3000 else (a250[k - 3000], b250[k - 3000])
            for fh, r in ((h1, ra), (h2, rb)):
                fh.write("@m%d\n%s\n+\n%s\n" % (k, "".join("ACGTN"[c] for c in r), "I" * len(r)))
    want = _sam("bwa-meme_mode3", prefix, [f1, f2], threads=8, chunk=700000)
    cmd = [os.path.join(REF, "bwa-meme_dropin"), "mem", "-7", "-Y", "-K", "700000", "-t", "8", prefix, f1, f2]
    r = subprocess.run(cmd, capture_output=True, env=dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_VERBOSE="1"), timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = [l for l in r.stdout.decode().split("\n") if not l.startswith("@PG")]
    assert len(got) == len(want) and len(want) > 12000
    diff = [(x, y) for x, y in zip(got, want) if x != y]
    assert not diff, "first differing SAM line:\n%s\n%s" % diff[0]
    err = r.stderr.decode()
    m = list(re.finditer(r"bwa_gen_cigar2 calls answered from the table (\d+), computed by the reference's function (\d+)", err))
    assert m and int(m[-1].group(1)) > 5000 and int(m[-1].group(1)) > 20 * int(m[-1].group(2)), err[-1500:]   # the CIGAR table answers (nearly) all calls
    assert re.search(r"0 reads chained on the host", err)
    m = list(re.finditer(r"SAM text on the device: (\d+) records formatted there so far .*?, (\d+) by the reference's mem_aln2sam", err))
    assert m and int(m[-1].group(1)) > 10000 and int(m[-1].group(1)) > 10 * int(m[-1].group(2)), err[-1500:]       # (only reads with supplementary records stay with the host)
    m = list(re.finditer(r"mate rescue on the device: (\d+) Smith-Waterman jobs posed.*took from the table (\d+), run by the reference's kernels (\d+)", err))
    assert m and int(m[-1].group(1)) > 200 and int(m[-1].group(2)) == int(m[-1].group(1)) and int(m[-1].group(3)) == 0, err[-1500:]
    # and with the CIGAR stage / the mate-rescue stage off the same SAM comes out (the tables only ever replace identical answers)
    got2 = _sam("bwa-meme_dropin", prefix, [f1, f2], env=dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_CIGAR="0", MEME_DROPIN_MATESW="0", MEME_DROPIN_SAM="0"), threads=8, chunk=700000)
    assert got2 == got


@pytest.mark.skipif(not (R.have("bwa-meme_dropin") and R.have("bwa-meme_mode3") and R.cpu_can_run()),
                    reason="compiled reference (oracle/_ref) not available on this box")
def test_sam_identical_with_smart_pairing_and_with_scores_beyond_the_seed_filter(tmp_path):
    """(1) `mem -p` (MEM_F_SMARTPE, src/fastmap.cpp:790-828): one interleaved file with a few unpaired reads in it; every chunk is split in
    two arrays of the aligner's own, each processed with a stack-local copy of the options -- nothing the binding's reader handed out, so
    nothing may run ahead of its turn (the chunk prefetcher of round 4 raced here).  Several chunks, with the binding's reader on and off.
    (2) -W with a match score beyond the device seed filter's 12-bit packing (199 x 25 >= 4096): the run falls back to the reference's
    per-batch extension functions instead of stopping in its first chunk."""
    g = synth.make_genome(400_000, seed=141, repeat_frac=0.06, n_families=5, n_dups=5, dup_len=1200)
    fa = str(tmp_path / "p.fa")
    synth.write_fasta(fa, g, contigs=3)
    prefix = build_index(fa, bits=14)
    a, b = _pe_reads(g, 3000, 150, 142, 0.01, 0.0015)
    rng = np.random.default_rng(143)
    fq = str(tmp_path / "inter.fq")
    with open(fq, "w") as fh:
        for k in range(a.shape[0]):
            mates = ((a[k], "/1"), (b[k], "/2")) if rng.random() > 0.04 else ((a[k], ""),)        # ~4 % singletons between the pairs
            for r, suf in mates:
                fh.write("@i%d%s\n%s\n+\n%s\n" % (k, suf, "".join("ACGTN"[c] for c in r), "I" * len(r)))
    for chunk in (100000000, 250000):
        want = _sam("bwa-meme_mode3", prefix, [fq], threads=8, chunk=chunk, opts=("-p",))
        for extra in ({}, {"MEME_DROPIN_IO": "0"}, {"MEME_DROPIN_VIRTUAL": "2"}):
            got = _sam("bwa-meme_dropin", prefix, [fq], env=dict(os.environ, MEME_INDEX_PREFIX=prefix, **extra), threads=8, chunk=chunk, opts=("-p",))
            assert len(got) == len(want) and len(want) > 5500
            diff = [(x, y) for x, y in zip(got, want) if x != y]
            assert not diff, "-p chunk=%d %r: first differing SAM line:\n%s\n%s" % ((chunk, extra) + diff[0])
    f1, f2 = str(tmp_path / "w1.fq"), str(tmp_path / "w2.fq")
    synth.write_fastq(f1, a[:1500], prefix="w")
    synth.write_fastq(f2, b[:1500], prefix="w")
    opts = ("-W", "5", "-A", "25", "-B", "60", "-O", "90", "-E", "20", "-L", "80", "-U", "200", "-T", "600")
    want = _sam("bwa-meme_mode3", prefix, [f1, f2], threads=8, opts=opts)
    err = []
    got = _sam("bwa-meme_dropin", prefix, [f1, f2], env=dict(os.environ, MEME_INDEX_PREFIX=prefix), threads=8, opts=opts, stderr=err)
    assert "beyond the device seed filter's limits" in err[0]
    assert len(got) == len(want) and len(want) > 3000
    diff = [(x, y) for x, y in zip(got, want) if x != y]
    assert not diff, "-A 25: first differing SAM line:\n%s\n%s" % diff[0]


@pytest.mark.skipif(not (R.have("bwa-meme_dropin") and R.have("bwa-meme_mode3") and R.cpu_can_run()),
                    reason="compiled reference (oracle/_ref) not available on this box")
def test_sam_identical_to_the_fm_index_path(tmp_path):
    """BASELINE configs[2] asks for the SAM diff against BWA-MEM2's FM-index path, not only against `mem -7`: the reference binary
    without -7 (FM-index SMEMs, its own index built by `index -a mem2`) on the same reads -- paired, 6 000 pairs."""
    g = synth.make_genome(400_000, seed=41, repeat_frac=0.08, n_families=6, n_dups=6, dup_len=1500)
    fa = str(tmp_path / "fm.fa")
    synth.write_fasta(fa, g, contigs=3)
    prefix = build_index(fa, bits=14)
    r = subprocess.run([os.path.join(REF, "bwa-meme_mode3"), "index", "-a", "mem2", prefix], capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    a, b = _pe_reads(g, 6000, 150, 45, 0.01, 0.0015)
    f1, f2 = str(tmp_path / "f1.fq"), str(tmp_path / "f2.fq")
    synth.write_fastq(f1, a, prefix="q")
    synth.write_fastq(f2, b, prefix="q")
    cmd = [os.path.join(REF, "bwa-meme_mode3"), "mem", "-Y", "-K", "100000000", "-t", "8", prefix, f1, f2]          # no -7: the FM-index engine
    rr = subprocess.run(cmd, capture_output=True, timeout=900)
    assert rr.returncode == 0, rr.stderr.decode()[-2000:]
    want = [l for l in rr.stdout.decode().split("\n") if not l.startswith("@PG")]
    got = _sam("bwa-meme_dropin", prefix, [f1, f2], env=dict(os.environ, MEME_INDEX_PREFIX=prefix), threads=8)
    assert len(got) == len(want) and len(want) > 12000
    diff = [(x, y) for x, y in zip(got, want) if x != y]
    assert not diff, "first differing SAM line:\n%s\n%s" % diff[0]
