#!/usr/bin/env python3
"""Mate-rescue kernel probe: python scripts/kswv_probe.py [jobs ...]  -> kernel ms, call ms and GCUPS per batch size."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from pymeme import hipapi
import oracle_py
from common import kswv_workload
base, ref, qer = kswv_workload(n=8000, seed=123, read_len=(150, 151))
want, cells = oracle_py.kswv_batch(base, ref, qer)
cpj = cells / base.shape[0]
ctx = hipapi.Context(0)
for n in [int(a) for a in sys.argv[1:]] or [30000, 200000]:
    jobs = np.concatenate([base] * ((n + base.shape[0] - 1) // base.shape[0]))[:n].copy()
    best = (1e9, 1e9)
    for it in range(4):
        t0 = time.perf_counter(); got, ms = ctx.kswv_batch_host(jobs.view(hipapi.KSWV_JOB), ref, qer); dt = time.perf_counter() - t0
        best = (min(best[0], ms), min(best[1], dt * 1e3))
    ok = np.array_equal(got[:min(n, base.shape[0])].view(np.int32), want[:min(n, base.shape[0])].view(np.int32))
    print("[kswv probe] %d jobs: kernel %.2f ms (%.0f GCUPS), call %.2f ms, %.2f M jobs/s; %.0f cells per job; matches the oracle: %s"
          % (n, best[0], cpj * n / best[0] / 1e6, best[1], n / best[1] / 1e3, cpj, ok), flush=True)
