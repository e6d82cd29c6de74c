"""Live re-check of the oracle against the COMPILED REFERENCE (oracle/_ref), on data that is not in the golden
set.  Runs wherever oracle/_ref exists and the CPU can execute it (build container and GPU box); skipped
otherwise.  CPU only."""
import os

import numpy as np
import pytest

import bsw_gen
import oracle_py as O
import ref_py as R
from common import build_index
from pymeme import synth

need_ref = pytest.mark.skipif(not (R.have("learned_seeding_mode3") and R.have("libbsw_ref.so") and R.cpu_can_run()),
                              reason="compiled reference (oracle/_ref) not available")


@need_ref
@pytest.mark.parametrize("seed,length,kw", [(301, 150, dict(exact_frac=0.3, n_frac=0.05)),
                                             (302, 100, dict(sub_rate=0.04, indel_rate=0.005, n_frac=0.1)),
                                             (303, 36, dict(sub_rate=0.0))])
def test_oracle_seeds_equal_live_reference(tmp_path, seed, length, kw):
    g = synth.make_genome(150_000, seed=seed, repeat_frac=0.1, n_families=4, n_dups=5, dup_len=1200)
    fa = str(tmp_path / "live.fa")
    synth.write_fasta(fa, g, contigs=2)
    prefix = build_index(fa, bits=13)
    reads, _, _ = synth.make_reads(g, 1500, length, seed=seed + 1000, **kw)
    fq = str(tmp_path / "live.fq")
    synth.write_fastq(fq, reads)
    want = R.run_seed_dump(prefix, fq, mode=3, timeout=300)
    off = np.arange(0, (reads.shape[0] + 1) * length, length, dtype=np.int64)
    sm, ns, hits, nh, _ = O.seed_batch(O.load_index_files(prefix), reads, off, smem_cap=256, hit_cap=1 << 13, threads=0)
    assert O.format_seed_dump(sm, ns, hits) == want


@need_ref
@pytest.mark.parametrize("kw", [dict(), dict(max_q=250, sub=0.08, indel=0.03), dict(max_q=30, h0_max=30)])
def test_oracle_bsw_equals_live_reference_scalar(kw):
    pairs, ref, qer = bsw_gen.make_pairs(3000, seed=77, **kw)
    for w, eb in ((100, 5), (13, 5), (200, 0)):
        prm = O.default_bsw_params(end_bonus=eb)
        mine = pairs.copy()
        O.bsw_batch(mine, ref, qer, w, prm, threads=0)
        theirs = R.bsw_run(0, pairs, ref, qer, w, prm)
        assert np.array_equal(bsw_gen.outputs(mine), bsw_gen.outputs(theirs))
