"""BASELINE configs[2] at a size where positions need more than 30 bits: a 512 Mbp genome (1.02 G suffixes, 8 reference sequences),
200 k read pairs through the reference aligner with the HIP backend bound in (seeding, chaining, extension and the CIGAR table on the
device) against the unmodified reference's `mem -7` on the same index files: SAM identical.  (The GRCh38-sized comparison is
bench.py's e2e leg; the FM-index engine is compared at 400 kbp in test_gpu_sam_e2e.py -- its index build takes minutes at this size.)"""
import os
import subprocess

import numpy as np
import pytest

import ref_py as R
from pymeme import hostapi, synth, workload

pytestmark = pytest.mark.gpu
os.environ.setdefault("MEME_DROPIN_SAM_CHECK", "1")
# round 6: mate rescue is posed on the device; in the tests the reference's own posing function runs over the same records as well and every job index / result is compared in the aligner
os.environ.setdefault("MEME_DROPIN_MATE_CHECK", "1")
os.environ.setdefault("MEME_DROPIN_MATESW", "1")          # (the opt-in mate-rescue stage too, whatever the number of jobs)
os.environ.setdefault("MEME_DROPIN_MATESW_MIN", "0")


@pytest.mark.skipif(not (R.have("bwa-meme_dropin") and R.have("bwa-meme_mode3") and R.cpu_can_run()),
                    reason="compiled reference (oracle/_ref) not available on this box")
def test_sam_identical_512mbp_paired(tmp_path):
    import psutil
    if psutil.virtual_memory().available < 80e9:
        pytest.skip("needs ~60 GB of host RAM for the index build and the reference's index expansion")
    root = "/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path)
    d = os.path.join(root, "meme_sam_scale_%d" % os.getpid())
    os.makedirs(d, exist_ok=True)
    try:
        l_pac = 512_000_000
        g = synth.make_genome(l_pac, seed=11)
        text, sa = hostapi.build_sa(g)
        l1, l2 = hostapi.train_prmi(text, sa)
        prefix = os.path.join(d, "ref.fa")
        hostapi.write_index(prefix, g, text, sa, l1, l2, n_contigs=8)
        del text, sa
        n = 200_000
        rng = np.random.default_rng(5)
        pos = rng.integers(0, l_pac - 700, size=n)
        ins = rng.integers(300, 500, size=n)
        ar = np.arange(150)

        def mut(x):
            sub = rng.random(x.shape) < 0.01
            return np.where(sub, (x + rng.integers(1, 4, size=x.shape, dtype=np.uint8)) & 3, x).astype(np.uint8)
        f1, f2 = os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")
        workload.write_fastq_fast(f1, mut(g[pos[:, None] + ar[None, :]]), prefix="p")
        workload.write_fastq_fast(f2, mut(3 - g[(pos + ins - 150)[:, None] + ar[None, :]][:, ::-1]), prefix="p")
        out = {}
        for exe, thr in (("bwa-meme_mode3", str(min(128, os.cpu_count() or 8))), ("bwa-meme_dropin", "32")):
            sam = os.path.join(d, exe + ".sam")
            with open(sam, "wb") as fh:
                r = subprocess.run([os.path.join(R.REF_DIR, exe), "mem", "-7", "-Y", "-K", "20000000", "-t", thr, prefix, f1, f2], stdout=fh, stderr=subprocess.PIPE,
                                   env=dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_VERBOSE="1"), timeout=1500)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            out[exe] = [l for l in open(sam, "rb") if not l.startswith(b"@PG")]
            if exe == "bwa-meme_dropin":
                assert b"0 reads chained on the host" in r.stderr
        a, b = out["bwa-meme_dropin"], out["bwa-meme_mode3"]
        assert len(a) == len(b) and len(b) > 2 * n
        diff = [(x, y) for x, y in zip(a, b) if x != y]
        assert not diff, "first differing SAM line:\n%s\n%s" % (diff[0][0].decode(), diff[0][1].decode())
        # BASELINE configs[2] names BWA-MEM2 as the yardstick: the reference WITHOUT -7 (FM-index SMEMs) on its own `index -a mem2` files of the
        # same 512 Mbp genome.  The index build takes the reference two minutes at this size (profiles/r04_fmi_512.log); MEME_TEST_FMI_512=0
        # leaves the leg out.
        if os.environ.get("MEME_TEST_FMI_512", "1") != "0":
            import time
            t0 = time.time()
            with open(prefix, "wb") as fh:                      # the genome as FASTA, the eight sequences the index above was written with
                alpha = np.frombuffer(b"ACGT", np.uint8)
                for c in range(8):
                    lo, hi = l_pac * c // 8, l_pac * (c + 1) // 8
                    seq = alpha[g[lo:hi]]
                    fh.write(b">chrS%d\n" % (c + 1))
                    rows = seq.shape[0] // 80
                    np.concatenate([seq[:rows * 80].reshape(rows, 80), np.full((rows, 1), 10, np.uint8)], axis=1).tofile(fh)
                    if seq.shape[0] > rows * 80:
                        fh.write(seq[rows * 80:].tobytes() + b"\n")
            r = subprocess.run([os.path.join(R.REF_DIR, "bwa-meme_mode3"), "index", "-a", "mem2", prefix], capture_output=True, timeout=3000)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            print("[fmi-512] `index -a mem2` of the 512 Mbp genome: %.0f s" % (time.time() - t0))
            sam = os.path.join(d, "fmi.sam")
            t0 = time.time()
            with open(sam, "wb") as fh:
                r = subprocess.run([os.path.join(R.REF_DIR, "bwa-meme_mode3"), "mem", "-Y", "-K", "20000000", "-t", str(min(128, os.cpu_count() or 8)), prefix, f1, f2],
                                   stdout=fh, stderr=subprocess.PIPE, timeout=3000)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            fmi = [l for l in open(sam, "rb") if not l.startswith(b"@PG")]
            diff = [(x, y) for x, y in zip(a, fmi) if x != y]
            print("[fmi-512] `mem` (FM-index engine) on %d pairs: %.0f s; SAM lines %d vs %d, differing %d" % (n, time.time() - t0, len(a), len(fmi), len(diff)))
            assert len(a) == len(fmi) and not diff, "first SAM line that differs from the FM-index engine's:\n%s\n%s" % (diff[0][0].decode(), diff[0][1].decode())
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)
