#!/usr/bin/env python3
"""Run-to-run stability of the bound aligner: the same input through oracle/_ref/bwa-meme_dropin several times per configuration, SAM files
compared line by line (first differing records printed).  python scripts/nondet_probe.py [Mbp] [Mpairs] [runs]   (the SAM-identity bar)"""
import os, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))
import numpy as np
from pymeme import hostapi, synth, workload
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 128
npairs = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 1000000
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
d = tempfile.mkdtemp(prefix="nondet_", dir="/dev/shm")
g = synth.make_genome(int(mbp * 1e6) & ~1, seed=11)
t0 = time.time(); text, sa = hostapi.build_sa(g); l1, l2 = hostapi.train_prmi(text, sa)
prefix = os.path.join(d, "ref.fa"); hostapi.write_index(prefix, g, text, sa, l1, l2, n_contigs=8); print("index %.1f s" % (time.time() - t0), flush=True)
rng = np.random.default_rng(5)
f1, f2 = os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")
for p0 in range(0, npairs, 1 << 20):
    m = min(1 << 20, npairs - p0)
    r1, r2 = workload.make_pairs_chunk(g, m, 150, rng, 0.01)
    workload.write_fastq_fast(f1, r1, prefix="p", first=p0, append=p0 > 0); workload.write_fastq_fast(f2, r2, prefix="p", first=p0, append=p0 > 0)
configs = [("default", {}), ("cigar0", {"MEME_DROPIN_CIGAR": "0"}), ("prefetch0", {"MEME_DROPIN_PREFETCH": "0"}), ("matesw0", {"MEME_DROPIN_MATESW": "0"})]
if os.environ.get("NONDET_CONFIGS"): configs = [c for c in configs if c[0] in os.environ["NONDET_CONFIGS"].split(",")]
ref = None
for name, extra in configs:
    for k in range(runs):
        out = os.path.join(d, "%s_%d.sam" % (name, k))
        env = dict(os.environ, MEME_INDEX_PREFIX=prefix, **extra)
        with open(out, "wb") as fh:
            r = subprocess.run([os.path.join(REPO, "oracle", "_ref", os.environ.get("NONDET_EXE", "bwa-meme_dropin")), "mem", "-7", "-Y", "-K", "100000000", "-t", "64", prefix, f1, f2], stdout=fh, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        lines = [l for l in open(out, "rb") if not l.startswith(b"@PG")]
        os.remove(out)
        if ref is None:
            ref = lines; print(name, k, "reference run,", len(lines), "lines", flush=True); continue
        diff = [(i, a, b) for i, (a, b) in enumerate(zip(ref, lines)) if a != b]
        print(name, k, "lines", len(lines), "differing", len(diff), flush=True)
        for i, a, b in diff[:6]:
            print("   line", i); print("   <", a.decode().rstrip()[:600]); print("   >", b.decode().rstrip()[:600])
import shutil; shutil.rmtree(d, ignore_errors=True)
