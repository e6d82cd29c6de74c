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
            for extra in ({}, {"MEME_DROPIN_PREFETCH": "0"}, {"MEME_DROPIN_VIRTUAL": "3"}, {"MEME_DROPIN_VERIFY": "1"}, {"MEME_DROPIN_VIRTUAL": "3", "MEME_DROPIN_VERIFY": "1", "MEME_DROPIN_PREFETCH": "0"}):
                got, err = _run("bwa-meme_dropin", prefix, fqs, threads, chunk, dict(base, **extra))
                assert len(got) == len(want[kname]) and len(got) > 2 * n
                diff = [(a, b) for a, b in zip(got, want[kname]) if a != b]
                assert not diff, "threads=%d -K %d %r: first differing SAM line\n%s\n%s" % ((threads, chunk, extra) + diff[0])
                n_runs += 1
                if extra.get("MEME_DROPIN_VERIFY"):
                    assert "VERIFY FAILED" not in err
                    vl = sorted(re.findall(r"verify chunk (-?\d+) (\S+) dev (\d+): (\d+) items, hash ([0-9a-f]+)", err))
                    stages = {v[1].split("-round")[0] for v in vl}
                    assert {"ext-records", "cigar", "mate-rescue", "sam-text"} <= stages, "stages verified: %r" % (stages,)
                    n_verify_lines += len(vl)
                    # the stage outputs themselves (not only the SAM) agree between runs that split the chunks the same way
                    key = (kname, extra.get("MEME_DROPIN_VIRTUAL", "1"))
                    assert hashes.setdefault(key, vl) == vl, "per-stage hashes differ between two runs of %r" % (key,)
    assert n_runs == 30 and n_verify_lines > 0
