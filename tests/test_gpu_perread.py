"""The reference's per-read seeding API on the backend (second row; reference src/LearnedIndex_seeding.h:207-297): the reference's
own harness source test/Learned_seeding_big_read.cpp -- the one BASELINE configs[1] names --, unmodified, linked in front of libbwa_pic.so with
bwa-meme_amd/binding/meme_perread.cpp's definitions of learned_index_load / Learned_getSMEMsAllPosOneThread[_step1only] /
Learned_bwtSeedStrategyAllPosOneThread[_mem_tradeoff] (oracle/_ref/learned_seeding_dropin).  Its seed dump must equal the golden dumps the
compiled reference made (tests/golden/g1_seeds_<L>.txt), and its SMEM totals after one, two and three rounds those of the reference harness."""
import os
import subprocess

import pytest

import ref_py as R
from common import GOLDEN, build_index

pytestmark = pytest.mark.gpu
EXE = os.path.join(R.REF_DIR, "learned_seeding_dropin")


def _run(exe, prefix, fq, threads, steps):
    r = subprocess.run([exe, prefix, fq, "1000", str(threads), str(steps)], capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout.decode()


@pytest.mark.skipif(not (R.have("learned_seeding_dropin") and R.cpu_can_run()), reason="oracle/_ref/learned_seeding_dropin not available on this box")
@pytest.mark.parametrize("length", [150, 250, 60, 25])
def test_reference_harness_on_the_backend_prints_the_golden_seed_dump(length):
    prefix = build_index(os.path.join(GOLDEN, "g1.fa"))
    fq = os.path.join(GOLDEN, "g1_reads_%d.fq" % length)
    want = open(os.path.join(GOLDEN, "g1_seeds_%d.txt" % length)).read()
    assert _run(EXE, prefix, fq, 1, 4) == want
    if length == 150:
        assert _run(EXE, prefix, fq, 3, 4) == want        # (three OpenMP threads, each with a ctx of its own; one batch, so the order is the file's)


@pytest.mark.skipif(not (R.have("learned_seeding_dropin") and R.have("learned_seeding_mode3") and R.cpu_can_run()), reason="compiled reference (oracle/_ref) not available on this box")
def test_smem_totals_per_round_equal_the_reference_harness():
    prefix = build_index(os.path.join(GOLDEN, "g1.fa"))
    fq = os.path.join(GOLDEN, "g1_reads_150.fq")
    for steps in (1, 2, 3):
        got = _run(EXE, prefix, fq, 2, steps).strip().split("\t")
        want = _run(os.path.join(R.REF_DIR, "learned_seeding_mode3"), prefix, fq, 2, steps).strip().split("\t")
        assert got[0] == want[0] == "[RESULT]" and got[4:] == want[4:], (steps, got, want)       # total SMEMs, reads
