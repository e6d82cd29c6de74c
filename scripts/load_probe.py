#!/usr/bin/env python3
"""Index-files-to-HBM probe: python scripts/load_probe.py [Mbp]
Builds an index on the device, writes the reference's files to /dev/shm and times meme_index_load_files in fresh processes with
several reader-thread counts (MEME_LOAD_THREADS; MEME_LOAD_TRACE prints the rate per file)."""
import os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))
if len(sys.argv) > 2 and sys.argv[1] == "--load":
    from pymeme import hipapi
    # PROBE_BLOCKING=1: the device's waits sleep instead of spinning (hipDeviceScheduleBlockingSync, what the binding sets before its
    # first ctx: INTEGRATION 2e) -- the loader waits on the GPU a few times per 16 MB piece
    if os.environ.get("PROBE_BLOCKING") == "1":
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        print("[load probe] hipSetDevice %d, hipSetDeviceFlags(blocking) %d" % (hip.hipSetDevice(0), hip.hipSetDeviceFlags(ctypes.c_uint(4))), flush=True)
    ctx = hipapi.Context(0)
    t0 = time.time(); ctx.load_index_files(sys.argv[2]); print("[load probe] threads %s, blocking waits %s: %.2f s" % (os.environ.get("MEME_LOAD_THREADS", "default"), os.environ.get("PROBE_BLOCKING", "0"), time.time() - t0), flush=True)
    sys.exit(0)
import numpy as np, torch
from pymeme import hipapi, hostapi, synth
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 1024
l_pac = int(mbp * 1e6) & ~1; n = 2 * l_pac
g = synth.make_genome(l_pac, seed=11)
ctx = hipapi.Context(0)
text = hipapi.fwd_rc_text(g)
d_text, d_sa = hipapi.build_sa_device(ctx, text)
d_pos5 = hipapi.pos5_from_sa_torch(ctx, d_sa, n)
sa = d_sa.cpu().numpy().view(np.uint64); del d_sa
d_pac, d_ent = hipapi.stage_entries_torch(ctx, n, d_text, d_pos5)
bits = 28 if 8.0 * n + 8 > 8.0e9 else 26 if 8.0 * n + 8 > 1.0e9 else 24
d_l2, n_l2, d_l1, n_l1 = hipapi.train_prmi_device(ctx, d_ent, n, bits)
l2 = d_l2.cpu().numpy().view(hostapi.RMI_DTYPE); l1 = d_l1.cpu().numpy().view(hostapi.RMI_DTYPE)[:n_l1]
del d_pac, d_ent, d_l2, d_l1, d_text, d_pos5; torch.cuda.empty_cache(); ctx.close()
os.makedirs("/dev/shm/loadprobe", exist_ok=True)
prefix = "/dev/shm/loadprobe/ref.fa"
hostapi.write_index(prefix, g, text, sa, l1, l2, n_contigs=4)
total = sum(os.path.getsize(prefix + e) for e in (".0123", ".pos_packed", ".suffixarray_uint64_L1_PARAMETERS", ".suffixarray_uint64_L2_PARAMETERS"))
print("[load probe] %.1f GB of index files" % (total / 1e9), flush=True)
for blocking in os.environ.get("PROBE_SEQUENCE", "0,1,0,1,0,1").split(","):
    env = dict(os.environ, PROBE_BLOCKING=blocking)
    if os.environ.get("PROBE_TRACE"): env["MEME_LOAD_TRACE"] = "1"
    subprocess.run([sys.executable, __file__, "--load", prefix], env=env)
import shutil; shutil.rmtree("/dev/shm/loadprobe")
