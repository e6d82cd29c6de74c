#!/usr/bin/env python3
"""End-to-end timing of the reference aligner with and without the HIP backend interposed
(oracle/_ref/bwa-meme_mode3 vs oracle/_ref/bwa-meme_dropin): python scripts/e2e_bench.py [Mbp] [Mpairs] [threads]
Prints wall time, reads/s and whether the SAM files are identical (minus @PG)."""
import hashlib, os, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from pymeme import hostapi, synth, workload
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 128
npairs = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 1000000
threads = int(sys.argv[3]) if len(sys.argv) > 3 else min(256, os.cpu_count() or 64)
log = lambda *a: print("[e2e]", *a, flush=True)
d = tempfile.mkdtemp(prefix="e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
g = synth.make_genome(int(mbp * 1e6) & ~1, seed=11)
t0 = time.time(); text, sa = hostapi.build_sa(g); l1, l2 = hostapi.train_prmi(text, sa)
prefix = os.path.join(d, "ref.fa"); hostapi.write_index(prefix, g, text, sa, l1, l2, n_contigs=8); log("index %.1f s" % (time.time() - t0))
rng = np.random.default_rng(5)
RL = int(os.environ.get("E2E_READ_LEN", "150")); SUB = float(os.environ.get("E2E_SUB", "0.01"))   # BASELINE configs[4] read class: 250 / 0.05
pos = rng.integers(0, g.shape[0] - 700 - RL, size=npairs); ins = rng.integers(300, 500, size=npairs) + (RL - 150)
ar = np.arange(RL)
def mut(x):
    sub = rng.random(x.shape) < SUB
    return np.where(sub, (x + rng.integers(1, 4, size=x.shape, dtype=np.uint8)) & 3, x).astype(np.uint8)
r1 = mut(g[pos[:, None] + ar[None, :]])
r2 = mut(3 - g[(pos + ins - RL)[:, None] + ar[None, :]][:, ::-1])
def fastq(reads, path):
    n, L = reads.shape
    names = np.char.add("@p", np.arange(n).astype(str)).astype("S")
    with open(path, "wb") as fh:
        alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
        seqs = alpha[reads]
        q = b"I" * L
        for i in range(n):
            fh.write(names[i] + b"\n" + seqs[i].tobytes() + b"\n+\n" + q + b"\n")
f1, f2 = os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")
t0 = time.time(); fastq(r1, f1); fastq(r2, f2); log("fastq %.1f s" % (time.time() - t0))
res = {}
runs = [("bwa-meme_mode3", None)] + [("bwa-meme_dropin", m) for m in os.environ.get("E2E_EXT_MODES", "device").split(",")]
if os.environ.get("E2E_REF_SWEEP"):              # the unmodified reference at several thread counts (its best is the honest baseline)
    runs = [("bwa-meme_mode3", "t" + t) for t in os.environ["E2E_REF_SWEEP"].split(",")]
if os.environ.get("E2E_SKIP_REF"): runs = runs[1:]
for exe, mode in runs:
    out = os.path.join(d, exe + ".sam")
    env = dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_VERBOSE="1")
    if not os.environ.get("E2E_NO_TUNABLES"):      # the allocator settings bench.py gives both binaries
        env.setdefault("GLIBC_TUNABLES", "glibc.malloc.tcache_count=4000:glibc.malloc.trim_threshold=1073741824:glibc.malloc.top_pad=67108864:glibc.malloc.mmap_threshold=33554432")
    nthr = threads
    if mode and mode.startswith("t"): nthr = int(mode[1:])
    elif mode: env["MEME_DROPIN_EXT"] = mode
    if os.environ.get("E2E_BSW_TRACE"): env["MEME_BSW_TRACE"] = "1"
    t0 = time.time()
    with open(out, "wb") as fh:
        chunk = os.environ.get("E2E_CHUNK", "100000000")       # E2E_CHUNK=default: the aligner's own chunking (10 M bases x threads)
        kopt = [] if chunk == "default" else ["-K", chunk]
        r = subprocess.run([os.path.join(REPO, "oracle", "_ref", exe), "mem", "-7", "-Y"] + kopt + ["-t", str(nthr), prefix, f1, f2], stdout=fh, stderr=subprocess.PIPE, env=env)
    wall = time.time() - t0
    err = r.stderr.decode()
    if os.environ.get("E2E_STDERR_DIR"):
        open(os.path.join(os.environ["E2E_STDERR_DIR"], exe + ".stderr"), "w").write(err)
    keys = [l for l in err.split("\n") if any(k in l for k in ("Runtime-build-index", "Total kernel", "LEARNED", "BSW time", "SAM Processing", "Overall time", "total time", "Loading", "meme-dropin", "meme bsw"))]
    h = hashlib.md5()
    nlines = 0
    with open(out, "rb") as fh:
        for line in fh:
            if not line.startswith(b"@PG"):
                h.update(line); nlines += 1
    res[exe + (":" + mode if mode else "")] = (wall, h.hexdigest(), nlines)
    log(exe, mode or "", "rc", r.returncode, "wall %.1f s" % wall, "->", "%.0f reads/s (wall, incl. index load)" % (2 * npairs / wall), "sam lines", nlines)
    for k in err.split("\n")[-45:]:
        if k.strip(): log("   ", k.strip())
log("SAM md5:", {k: v[1] for k, v in res.items()}, "all identical:", len({v[1] for v in res.values()}) == 1)
import shutil; shutil.rmtree(d, ignore_errors=True)
