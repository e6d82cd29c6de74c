#!/usr/bin/env python3
"""The compiled reference's seeding harness (oracle/_ref/learned_seeding_mode3 = test/Learned_seeding_big_read.cpp, MODE 3, AVX-512)
at several thread counts on the same index and reads: python scripts/ref_seed_sweep.py [Mbp] [Mreads] [threads,threads,...]
-- which thread count is the reference's best on this box (bench.py's cpu_baseline quotes that one)."""
import os, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from pymeme import hostapi, synth, workload
import oracle_py as O
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 512
nreads = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 2000000
threads = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "32,64,128,256").split(",")]
d = tempfile.mkdtemp(prefix="rss_", dir="/dev/shm")
g = synth.make_genome(int(mbp * 1e6) & ~1, seed=11)
t0 = time.time(); text, sa = hostapi.build_sa(g); l1, l2 = hostapi.train_prmi(text, sa)
prefix = os.path.join(d, "ref.fa"); hostapi.write_index(prefix, g, text, sa, l1, l2, n_contigs=8); print("[sweep] index %.1f s" % (time.time() - t0), flush=True)
reads = workload.make_reads_fast(g, nreads, 150, seed=1000)
alpha = np.frombuffer(b"ACGTN", np.uint8)
fq = os.path.join(d, "s.fq")
with open(fq, "wb") as fh:
    q = b"I" * 150
    for i in range(nreads):
        fh.write(b"@r%d\n" % i + alpha[reads[i]].tobytes() + b"\n+\n" + q + b"\n")
hz = O.tsc_hz()
for t in threads:
    r = subprocess.run([os.path.join(REPO, "oracle", "_ref", "learned_seeding_mode3"), prefix, fq, "1000", str(t), "3"], capture_output=True, text=True,
                       env=dict(os.environ, OMP_NUM_THREADS=str(t)))
    cyc = [float(l.split()[1]) for l in r.stderr.splitlines() if l.startswith("Consumed:")]
    print("[sweep] %3d threads: %s" % (t, "%.3f s -> %.0f reads/s" % (cyc[0] / hz, nreads / (cyc[0] / hz)) if cyc else "failed: " + r.stderr[-300:]), flush=True)
import shutil; shutil.rmtree(d, ignore_errors=True)
