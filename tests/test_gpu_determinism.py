"""Run-to-run and arrangement-to-arrangement identity of the bound aligner's output (round 6; SURVEY 8: SAM diff == empty; the pipeline whose
ordering the hooks must preserve: reference src/fastmap.cpp:730-866).  The same paired input through oracle/_ref/bwa-meme_dropin under varied
worker-thread counts, -K chunk sizes, the next chunk ahead of its turn on / off, virtual device slots and MEME_DROPIN_VERIFY=1 -- every device
stage of every chunk twice on two ctxs of the GPU, outputs compared byte for byte inside the aligner, a difference is fatal -- and every SAM
equal to the unmodified reference's for the same -K.  scripts/r06_soak.py is the long form (hundreds of runs at 128 Mbp + ThreadSanitizer
runs; profiles/r06_determinism.md)."""
import os
import re
import subprocess

import numpy as np
import pytest

import ref_py as R
from common import build_index
from pymeme import synth

pytestmark = pytest.mark.gpu
REF = R.REF_DIR


def _run(exe, prefix, fqs, threads, chunk, env):
    cmd = [os.path.join(REF, exe), "mem", "-7", "-Y", "-K", str(chunk), "-t", str(threads), prefix] + fqs
    r = subprocess.run(cmd, capture_output=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return [l for l in r.stdout.decode().split("\n") if not l.startswith("@PG")], r.stderr.decode()


@pytest.mark.skipif(not (R.have("bwa-meme_dropin") and R.have("bwa-meme_mode3") and R.cpu_can_run()), reason="compiled reference (oracle/_ref) not available on this box")
def test_sam_identical_under_varied_arrangements_and_stage_verify(tmp_path):
    g = synth.make_genome(2_000_000, seed=141, repeat_frac=0.10, n_families=8, n_dups=8, dup_len=1500, poly_runs=2)
    fa = str(tmp_path / "det.fa")
    synth.write_fasta(fa, g, contigs=4)
    prefix = build_index(fa, bits=16)
    n = 20000
    r1, pos, _ = synth.make_reads(g, n, 150, seed=142, n_frac=0.02, exact_frac=0.2)
    rng = np.random.default_rng(143)
    p2 = np.clip(pos + rng.integers(300, 500, size=n) - 150, 0, g.shape[0] - 160)
    r2 = 3 - g[p2[:, None] + np.arange(150)[None, :]][:, ::-1]
    r2 = np.where(rng.random(r2.shape) < 0.015, (r2 + 1) & 3, r2).astype(np.uint8)
    fqs = [str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")]
    synth.write_fastq(fqs[0], r1, prefix="p")
    synth.write_fastq(fqs[1], r2, prefix="p")
    base = dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_MATESW="1", MEME_DROPIN_MATESW_MIN="0")
    base.pop("MEME_DROPIN_SAM_CHECK", None)
    chunks = {"big": 100000000, "small": 700000}                      # one chunk; nine chunks, the last one ragged
    want = {k: _run("bwa-meme_mode3", prefix, fqs, 8, c, dict(os.environ))[0] for k, c in chunks.items()}
    n_runs = n_verify_lines = 0
    hashes = {}
    for threads in (4, 16, 64):
        for kname, chunk in chunks.items():
            for extra in ({}, {"MEME_DROPIN_PREFETCH": "0"}, {"MEME_DROPIN_VIRTUAL": "3"}, {"MEME_DROPIN_VERIFY": "1"}, {"MEME_DROPIN_VIRTUAL": "3", "MEME_DROPIN_VERIFY": "1", "MEME_DROPIN_PREFETCH": "0"},
                          {"MEME_DROPIN_HALVES": "0", "MEME_DROPIN_MATE_POSE": "0"}):      # (the SAM phase in one piece, mate rescue posed on the host: round 5's arrangement)
                got, err = _run("bwa-meme_dropin", prefix, fqs, threads, chunk, dict(base, **extra))
                assert len(got) == len(want[kname]) and len(got) > 2 * n
                diff = [(a, b) for a, b in zip(got, want[kname]) if a != b]
                assert not diff, "threads=%d -K %d %r: first differing SAM line\n%s\n%s" % ((threads, chunk, extra) + diff[0])
                n_runs += 1
                if extra.get("MEME_DROPIN_VERIFY"):
                    assert "VERIFY FAILED" not in err
                    vl = sorted(re.findall(r"verify chunk (-?\d+) (\S+) dev (\d+): (\d+) items, hash ([0-9a-f]+)", err))
                    stages = {v[1].decode().split("-round")[0] if isinstance(v[1], bytes) else v[1].split("-round")[0] for v in vl}
                    assert {"ext-records", "cigar", "mate-rescue", "sam-text"} <= stages, "stages verified: %r" % (stages,)
                    n_verify_lines += len(vl)
                    # the stage outputs themselves (not only the SAM) agree between runs that split the chunks the same way
                    key = (kname, extra.get("MEME_DROPIN_VIRTUAL", "1"))
                    assert hashes.setdefault(key, vl) == vl, "per-stage hashes differ between two runs of %r" % (key,)
    assert n_runs == 36 and n_verify_lines > 0


@pytest.mark.skipif(not (R.have("bwa-meme_dropin") and R.have("bwa-meme_mode3") and R.cpu_can_run()), reason="compiled reference (oracle/_ref) not available on this box")
def test_read_counter_race_of_the_reference_pipeline_is_pinned(tmp_path):
    """The cause of round 5's one differing SAM md5 (found by ThreadSanitizer, profiles/r06_determinism.md): the reference's output step adds a chunk's
    reads to aux->n_processed (src/fastmap.cpp:845) while the next chunk's process step reads it (:832) -- unordered; the value seeds the tie-breaking hash
    of equally good alignments and pairs (src/bwamem.cpp:2010, src/bwamem_pair.cpp:412).  MEME_DROPIN_TEST_OUTPUT_DELAY_MS forces the rare order (the output
    step late): the binding, which counts the reads itself, must still print the reference's SAM; with MEME_DROPIN_NPROC=ref (the counter as the pipeline
    read it) the same run shows what the race does -- the same number of lines, other placements of reads that have two equally good ones."""
    g = synth.make_genome(1_500_000, seed=151, repeat_frac=0.05, n_families=6, n_dups=12, dup_len=2500)      # exact duplicates: tied placements
    fa = str(tmp_path / "race.fa")
    synth.write_fasta(fa, g, contigs=2)
    prefix = build_index(fa, bits=16)
    n = 16000
    r1, pos, _ = synth.make_reads(g, n, 150, seed=152, n_frac=0.0, exact_frac=0.5)
    rng = np.random.default_rng(153)
    p2 = np.clip(pos + rng.integers(300, 500, size=n) - 150, 0, g.shape[0] - 160)
    r2 = (3 - g[p2[:, None] + np.arange(150)[None, :]][:, ::-1]).astype(np.uint8)
    fqs = [str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")]
    synth.write_fastq(fqs[0], r1, prefix="p")
    synth.write_fastq(fqs[1], r2, prefix="p")
    chunk = 600000                                                   # eight chunks
    want, _ = _run("bwa-meme_mode3", prefix, fqs, 8, chunk, dict(os.environ))
    base = dict(os.environ, MEME_INDEX_PREFIX=prefix)
    got, err = _run("bwa-meme_dropin", prefix, fqs, 8, chunk, dict(base, MEME_DROPIN_TEST_OUTPUT_DELAY_MS="120"))
    assert "the ordered count is used" in err, "the delay did not produce the stale counter:\n" + err[-1500:]
    assert got == want, "the binding's output depends on the order of the pipeline's two threads"
    racy, err = _run("bwa-meme_dropin", prefix, fqs, 8, chunk, dict(base, MEME_DROPIN_TEST_OUTPUT_DELAY_MS="120", MEME_DROPIN_NPROC="ref"))
    assert "used as given" in err
    ndiff = sum(1 for a, b in zip(racy, want) if a != b)
    assert len(racy) == len(want) and ndiff > 0, "expected the stale counter to move tied placements (it is what the reference's own race does); %d differing lines" % ndiff
    # ... and only reads with equally good alternatives move: every differing record is a mapping-quality-0..few record or its mate
    print("stale read counter: %d of %d SAM lines differ" % (ndiff, len(want)))
