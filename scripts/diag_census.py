#!/usr/bin/env python3
"""Census of the searches a "diagonal + plcp" shortcut could take off the SA-search kernel (round-4 review, item 3), on the oracle (CPU):
python scripts/diag_census.py [Mbp] [reads].  The genome and the reads are the benchmark's (synth.make_genome seed 11, workload.make_reads_fast
seed 1000) at a size a host can index; the instrumented oracle (oracle/meme_oracle.c census_search) counts per round the searches whose answer
follows from a unique locus the read has already met, and checks every such claim against the search's real answer."""
import ctypes as C
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_py as O
from pymeme import hostapi, synth, workload

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 64
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
g = synth.make_genome(int(mbp * 1e6) & ~1, seed=11)
t0 = time.time()
text, sa = hostapi.build_sa(g)
n = text.shape[0]
print("genome %.0f Mbp, %d suffixes, suffix array in %.0f s" % (mbp, n, time.time() - t0), flush=True)
L = O.lib()
plcp = np.zeros(n, np.uint8)
L.orc_build_plcp(C.c_void_p(text.ctypes.data), C.c_void_p(sa.ctypes.data), C.c_int64(n), C.c_void_p(plcp.ctypes.data))
print("plcp: mean %.2f, share >= 19: %.4f, saturated: %.6f" % (plcp.mean(), (plcp >= 19).mean(), (plcp == 255).mean()), flush=True)
reads = workload.make_reads_fast(g, nreads, 150, seed=1000)
off = np.arange(0, (nreads + 1) * 150, 150, dtype=np.int64)
idx = O.Index(text, sa)
L.orc_diag_census_enable(C.c_void_p(plcp.ctypes.data))
O.seed_batch(idx, reads, off, smem_cap=512, hit_cap=8192, threads=0)
out = (C.c_longlong * 24)()
L.orc_diag_census_get(out)
L.orc_diag_census_enable(C.c_void_p(0))
rows = np.array(list(out), dtype=np.int64).reshape(4, 6)
names = ["round 1 (SMEM zig-zag)", "re-seeding (round 2)", "third round"]
tot = rows[:3, 0].sum()
print("| round | searches / read | answerable from an earlier unique SMEM | ... from any earlier unique search result | claims that were wrong | of (b): L_d < min_seed_len |")
print("|---|---|---|---|---|---|")
for k in range(3):
    r = rows[k]
    print("| %s | %.2f | %.2f (%.1f %%) | %.2f (%.1f %%) | %d | %.2f |" % (names[k], r[0] / nreads, r[1] / nreads, 100.0 * r[1] / max(r[0], 1), r[2] / nreads, 100.0 * r[2] / max(r[0], 1), r[3], r[4] / nreads))
print("| all | %.2f | %.2f (%.1f %%) | %.2f (%.1f %%) | %d | |" % (tot / nreads, rows[:3, 1].sum() / nreads, 100.0 * rows[:3, 1].sum() / tot, rows[:3, 2].sum() / nreads,
                                                              100.0 * rows[:3, 2].sum() / tot, rows[:3, 3].sum()))
