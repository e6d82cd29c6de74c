"""Every stage of the path on a REPEAT-DENSE genome (round 6): 42 % interspersed 300-bp repeats at 12 % divergence, tandem
satellites, exact duplications -- the workload on which the overflow tiers of seeding, the max_occ subsampling of hit lists
(reference src/bwamem.cpp:1154-1160), the wavefront-per-read and B-tree chaining tiers and the many-jobs-per-read side of the extension carry
load (the benchmark's headline genome is 98 % unique).  Seeds against orc_seed_batch, chains against orc_chain_read (every read), alignment
records against orc_extend_read, and the bound aligner's SAM against the unmodified reference's on a smaller genome of the same recipe."""
import os
import subprocess

import numpy as np
import pytest

import oracle_py as O
import ref_py as R
from common import build_index
from pymeme import hipapi, synth, workload

pytestmark = pytest.mark.gpu


def test_seeding_chaining_extension_equal_the_oracle_on_a_repeat_dense_genome():
    import torch
    l_pac = 64_000_000
    n = 2 * l_pac
    g = workload.repeat_dense_genome(l_pac)
    text = hipapi.fwd_rc_text(g)
    ctx = hipapi.Context(0)
    try:
        d_text, d_sa = hipapi.build_sa_device(ctx, text)
        d_pos5 = hipapi.pos5_from_sa_torch(ctx, d_sa, n)
        sa = d_sa.cpu().numpy().view(np.uint64)
        del d_sa
        torch.cuda.empty_cache()
        d_pac, d_ent = hipapi.stage_entries_torch(ctx, n, d_text, d_pos5)
        d_l2, n_l2, d_l1, n_l1 = hipapi.train_prmi_device(ctx, d_ent, n, 24)
        keep = (d_pac, d_ent) + hipapi.attach_index_torch(ctx, n, d_pac, d_ent, d_l2, n_l2, d_l1, n_l1)
        idx = O.Index(text, sa)
        nreads, L = 6000, 150
        reads = workload.make_reads_fast(g, nreads, L, seed=79)
        off = np.arange(0, (nreads + 1) * L, L, dtype=np.int64)
        # ---- seeding: every SMEM, every hit position
        smems, so, hits, ho = ctx.seed_batch(reads, off)
        slots, counts, hl = hipapi.smems_to_slots(smems, so, hits, ho)
        sm, ns, oh, nh, _ = O.seed_batch(idx, reads, off, smem_cap=4096, hit_cap=1 << 18, threads=0)
        assert O.format_seed_dump(slots, counts, hl) == O.format_seed_dump(sm, ns, oh)
        hits_per_read = hits.shape[0] / nreads
        assert hits_per_read > 25 and int(smems["hitcount"].max()) > 500, ("the genome is not repeat-dense for these reads: %.1f hits per read, longest hit list %d (max_occ 500)"
                                                                            % (hits_per_read, int(smems["hitcount"].max())))
        # the same with 8 SMEM slots per read in the first tier: most reads go through the overflow tiers
        ctx.set_tuning("smem_cap", 8)
        smems2, so2, hits2, ho2 = ctx.seed_batch(reads, off)
        ctx.set_tuning("smem_cap", 128)
        s2, c2, h2 = hipapi.smems_to_slots(smems2, so2, hits2, ho2)
        assert O.format_seed_dump(s2, c2, h2) == O.format_seed_dump(sm, ns, oh)
        # ---- chaining of the batch where its seeds lie: every read against the oracle (B-tree, max_occ subsampling, the filter)
        smems, so, hits, ho = ctx.seed_batch_host(reads.reshape(-1), off)
        contigs = [(l_pac * i // 4, l_pac * (i + 1) // 4 - l_pac * i // 4, 0) for i in range(4)]
        copt = hipapi.default_chain_opt(l_pac)
        res = ctx.chain_last_batch_host(contigs, copt)
        tm = ctx.timings()
        n_bad, first_bad = O.chain_compare_batch(smems, so, hits, ho, np.full(nreads, L, np.int32), np.array([c[0] for c in contigs], np.int64), np.zeros(4, np.uint8),
                                                 O.default_chain_opt(l_pac), res)
        assert n_bad == 0 and res["n_fallback"] == 0, (n_bad, first_bad)
        assert res["n_tier2"] > nreads // 20, "the wavefront-per-read chaining tiers carry no load: %d of %d reads" % (res["n_tier2"], nreads)
        # ---- extension: every alignment record against the oracle's restatement of mem_chain2aln_across_reads_V2
        Rr = ctx.extend_last_batch_host(contigs, copt)
        ch = ctx.chain_last_batch_host(contigs, copt)
        want, (jobs, retried) = O.extend_batch(reads.reshape(-1), off, ch["chain_off"], O.chains_as_orc(ch["chains"]), ch["seed_off"], ch["seeds"], ch["frac_rep"], text, l_pac,
                                               np.array([c[0] for c in contigs], np.int64), np.array([c[1] for c in contigs], np.int32))
        assert np.array_equal(Rr["reg_off"], ch["seed_off"])
        for f in O.ALNREG_FIELDS:
            assert np.array_equal(Rr["regs"][f].astype(np.int64), want[f].astype(np.int64)), f
        assert Rr["n_pairs"] / nreads > 8, "extension jobs per read: %.1f" % (Rr["n_pairs"] / nreads)
        # ... and in rounds (what the bound aligner asks for): the surviving records, byte for byte
        ctx.set_tuning("ext_live_only", 1)
        RL = ctx.extend_last_batch_host(contigs, copt)
        ctx.set_tuning("ext_live_only", 0)
        keep_m = Rr["regs"]["qe"] > Rr["regs"]["qb"]
        assert hipapi.records_equal(RL["regs"], Rr["regs"][keep_m])
        print("repeat-dense: %.1f hits / read, %d of %d reads in the wavefront chaining tiers (%d through the B-tree tier), %.1f extension jobs / read"
              % (hits_per_read, res["n_tier2"], nreads, tm.chain_tier3_reads, Rr["n_pairs"] / nreads))
    finally:
        ctx.close()


@pytest.mark.skipif(not (R.have("bwa-meme_dropin") and R.have("bwa-meme_mode3") and R.cpu_can_run()), reason="compiled reference (oracle/_ref) not available on this box")
def test_sam_identical_on_a_repeat_dense_genome(tmp_path):
    g = workload.repeat_dense_genome(6_000_000, seed=81)
    fa = str(tmp_path / "rd.fa")
    synth.write_fasta(fa, g, contigs=3)
    prefix = build_index(fa, bits=18, threads=min(32, os.cpu_count() or 4))
    n = 12000
    rng = np.random.default_rng(82)
    r1, r2 = workload.make_pairs_chunk(g, n, 150, rng, 0.01, 0.002)
    fqs = [str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")]
    workload.write_fastq_fast(fqs[0], r1, prefix="p")
    workload.write_fastq_fast(fqs[1], r2, prefix="p")

    def sam(exe, env, chunk):
        r = subprocess.run([os.path.join(R.REF_DIR, exe), "mem", "-7", "-Y", "-K", str(chunk), "-t", "16", prefix] + fqs, capture_output=True, env=env, timeout=1500)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        return [l for l in r.stdout.decode().split("\n") if not l.startswith("@PG")]

    for chunk in (100000000, 900000):
        want = sam("bwa-meme_mode3", dict(os.environ), chunk)
        got = sam("bwa-meme_dropin", dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_MATESW="1", MEME_DROPIN_MATESW_MIN="0", MEME_DROPIN_SAM_CHECK="1", MEME_DROPIN_VERIFY="1"), chunk)
        assert len(got) == len(want) and len(want) > 2 * n
        diff = [(a, b) for a, b in zip(got, want) if a != b]
        assert not diff, "-K %d: first differing SAM line\n%s\n%s" % ((chunk,) + diff[0])
