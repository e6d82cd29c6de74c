#!/usr/bin/env python3
"""Latency of small BSW calls through the host-buffer ABI (what the reference's call sites issue): bsw_latency.py [npairs]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))
import numpy as np
from pymeme import hipapi, workload
ctx = hipapi.Context(0)
for n in [int(a) for a in sys.argv[1:]] or [200, 2000, 20000]:
    pairs, ref, qer, base = workload.make_bsw_pairs(n, seed=7, read_len=150, base=n)
    for it in range(3): ctx.bsw_batch(pairs.copy(), ref, qer, 100)
    t0 = time.perf_counter()
    reps = 50
    for it in range(reps): ctx.bsw_batch(pairs, ref, qer, 100)
    dt = (time.perf_counter() - t0) / reps
    print("[bsw latency] %d pairs per call: %.3f ms per call (kernel part %.3f ms) -> %.2f M pairs/s" % (n, dt * 1e3, ctx.timings().bsw_kernel_ms, n / dt / 1e6), flush=True)
