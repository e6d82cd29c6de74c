#!/usr/bin/env python3
"""Round 6, the determinism question (bit-identical SAM; reference pipeline src/fastmap.cpp:730-866): the same paired input through
oracle/_ref/bwa-meme_dropin many times under VARIED arrangements -- worker threads, -K chunk size, the next chunk ahead of its turn on / off,
virtual device slots, the SAM-phase stages on / off, MEME_DROPIN_VERIFY (every device stage twice on two ctxs, compared in the aligner) -- and
every SAM compared with the first run's line by line.  -K changes the chunking and with it mem_pestat's per-chunk statistics, i.e. the
reference's own output: runs are compared within their -K class, and each class's first run against the unmodified reference binary.

  python scripts/r06_soak.py [Mbp] [Mpairs] [runs] [out.json]        (env SOAK_TSAN=k: k extra runs of bwa-meme_dropin_tsan, log kept)

Prints one line per run and writes a JSON summary (runs, configurations, differing runs, verify lines seen, per-stage hash agreement)."""
import hashlib, json, os, re, shutil, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))
import numpy as np
from pymeme import hostapi, synth, workload

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 128
npairs = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 300000
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 200
out_json = sys.argv[4] if len(sys.argv) > 4 else os.path.join(REPO, "gpurun_out", "r06_soak.json")
REFD = os.path.join(REPO, "oracle", "_ref")
d = tempfile.mkdtemp(prefix="soak_", dir="/dev/shm")
g = synth.make_genome(int(mbp * 1e6) & ~1, seed=11, repeat_frac=0.05, n_families=12, n_dups=8, dup_len=2000)
t0 = time.time()
text, sa = hostapi.build_sa(g)
l1, l2 = hostapi.train_prmi(text, sa)
prefix = os.path.join(d, "ref.fa")
hostapi.write_index(prefix, g, text, sa, l1, l2, n_contigs=8)
del text, sa
print("index of %.0f Mbp in %.1f s" % (mbp, time.time() - t0), flush=True)
rng = np.random.default_rng(5)
f1, f2 = os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")
for p0 in range(0, npairs, 1 << 20):
    m = min(1 << 20, npairs - p0)
    # a tenth of the pairs from the 250-bp / 5 % class would change the read length inside a file: keep one class per file, errors mixed instead
    r1, r2 = workload.make_pairs_chunk(g, m, 150, rng, 0.02 if (p0 >> 20) & 1 else 0.01, 0.002)
    workload.write_fastq_fast(f1, r1, prefix="p", first=p0, append=p0 > 0)
    workload.write_fastq_fast(f2, r2, prefix="p", first=p0, append=p0 > 0)
bases = npairs * 300
K_small = max(400000, (bases // 7) // 300 * 300 + 150)          # ~7 chunks, the last one ragged
K_CLASSES = {"K100M": 100000000, "Ksmall": K_small}

def variants():
    """An endless, deterministic cycle of arrangements."""
    k = 0
    while True:
        for threads in (4, 16, 64):
            for kname in ("Ksmall", "K100M"):
                for extra in ({}, {"MEME_DROPIN_PREFETCH": "0"}, {"MEME_DROPIN_VIRTUAL": "3"}, {"MEME_DROPIN_VERIFY": "1"}, {"MEME_DROPIN_VIRTUAL": "2", "MEME_DROPIN_VERIFY": "1"},
                              {"MEME_DROPIN_SAM": "0"}, {"MEME_DROPIN_CIGAR": "0", "MEME_DROPIN_MATESW": "0"}, {"MEME_DROPIN_PREPASS_OVERLAP": "0"}, {"MEME_DROPIN_VIRTUAL": "8"},
                              {"MEME_DROPIN_HALVES": "0"}, {"MEME_DROPIN_MATE_POSE": "0"}, {"MEME_DROPIN_HALVES": "0", "MEME_DROPIN_VERIFY": "1"}, {"MEME_DROPIN_MATE_CHECK": "1", "MEME_DROPIN_VIRTUAL": "3"}):
                    yield k, threads, kname, dict(extra)
                    k += 1

def run(exe, threads, K, extra, stderr_path=None, timeout=1800):
    env = dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_MATESW_MIN="0", **extra)
    cmd = [os.path.join(REFD, exe), "mem", "-7", "-Y", "-K", str(K), "-t", str(threads), prefix, f1, f2]
    if exe.endswith("_tsan") and shutil.which("setarch"):
        cmd = ["setarch", "x86_64", "-R"] + cmd          # (gcc 11's ThreadSanitizer runtime and 32 bits of mmap entropy do not go together)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=timeout)
    if stderr_path:
        open(stderr_path, "wb").write(p.stderr[-200000:])
    assert p.returncode == 0 and b"FATAL: ThreadSanitizer" not in p.stderr, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG")]
    vl = sorted(re.findall(rb"verify chunk (-?\d+) (\S+) dev (\d+): (\d+) items, hash ([0-9a-f]+)", p.stderr))
    global last_stale
    last_stale = max([int(x) for x in re.findall(rb"(\d+) chunk\(s\) so far", p.stderr)] or [0])     # chunks whose pipeline read counter was stale (the reference's own race)
    return lines, vl

last_stale = 0
first, verify_ref, summary = {}, {}, {"mbp": mbp, "pairs": npairs, "K": K_CLASSES, "runs": [], "differing_runs": 0, "verify_lines": 0, "verify_hash_mismatch": 0}
# the unmodified reference per -K class (what "identical" is measured against)
ref_lines = {}
for kname, K in K_CLASSES.items():
    t = time.time()
    env = dict(os.environ)
    p = subprocess.run([os.path.join(REFD, "bwa-meme_mode3"), "mem", "-7", "-Y", "-K", str(K), "-t", "32", prefix, f1, f2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=3600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    ref_lines[kname] = [l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG")]
    print("reference %s: %d lines, md5 %s, %.1f s" % (kname, len(ref_lines[kname]), hashlib.md5(b"\n".join(ref_lines[kname])).hexdigest()[:12], time.time() - t), flush=True)
t_start = time.time()
for k, threads, kname, extra in variants():
    if k >= runs:
        break
    t = time.time()
    lines, vl = run("bwa-meme_dropin", threads, K_CLASSES[kname], extra)
    ref = ref_lines[kname]
    ndiff = sum(1 for a, b in zip(ref, lines) if a != b) + abs(len(ref) - len(lines))
    # per-stage hashes must agree between runs that split the chunk the same way (same -K, same number of device slots)
    split = (kname, extra.get("MEME_DROPIN_VIRTUAL", "1"), extra.get("MEME_DROPIN_HALVES", "1"))
    hm = 0
    if vl:
        if split not in verify_ref:
            verify_ref[split] = vl
        else:
            hm = sum(1 for a, b in zip(verify_ref[split], vl) if a != b) + abs(len(verify_ref[split]) - len(vl))
    rec = {"run": k, "threads": threads, "K": kname, "env": extra, "lines": len(lines), "differing_lines": ndiff, "verify_lines": len(vl), "verify_hash_mismatch": hm, "stale_counter_chunks": last_stale, "s": round(time.time() - t, 2)}
    summary["runs"].append(rec)
    summary["differing_runs"] += ndiff > 0
    summary["verify_lines"] += len(vl)
    summary["verify_hash_mismatch"] += hm
    summary["stale_counter_chunks"] = summary.get("stale_counter_chunks", 0) + last_stale
    print(json.dumps(rec), flush=True)
    if ndiff:
        bad = [(i, a, b) for i, (a, b) in enumerate(zip(ref, lines)) if a != b][:4]
        for i, a, b in bad:
            print("   line", i); print("   <", a.decode()[:500]); print("   >", b.decode()[:500])
summary["soak_s"] = round(time.time() - t_start, 1)
# ThreadSanitizer runs of the same workload (log kept next to the summary)
n_tsan = int(os.environ.get("SOAK_TSAN", "0"))
tsan = []
for k in range(n_tsan):
    threads, kname, extra = [(16, "Ksmall", {}), (16, "Ksmall", {"MEME_DROPIN_VIRTUAL": "2"}), (8, "K100M", {}), (16, "Ksmall", {"MEME_DROPIN_PREFETCH": "0"})][k % 4]
    log = os.path.join(os.path.dirname(out_json), "r06_tsan_%d" % k)
    env = dict(extra, TSAN_OPTIONS="suppressions=%s log_path=%s exitcode=0 history_size=4 second_deadlock_stack=1 report_signal_unsafe=0" % (os.path.join(REPO, "oracle", "tsan.supp"), log))
    t = time.time()
    try:
        lines, _ = run("bwa-meme_dropin_tsan", threads, K_CLASSES[kname], env, stderr_path=log + ".stderr", timeout=2400)
        ref = ref_lines[kname]
        ndiff = sum(1 for a, b in zip(ref, lines) if a != b) + abs(len(ref) - len(lines))
        reports = 0
        for fn in os.listdir(os.path.dirname(log)):
            if fn.startswith(os.path.basename(log) + "."):
                reports += open(os.path.join(os.path.dirname(log), fn), errors="replace").read().count("WARNING: ThreadSanitizer")
        tsan.append({"run": k, "threads": threads, "K": kname, "env": extra, "differing_lines": ndiff, "tsan_reports": reports, "s": round(time.time() - t, 1)})
    except Exception as e:                       # (the tool and the HIP runtime in one process: a failure to start is a result too)
        tsan.append({"run": k, "error": str(e)[-1500:]})
    print("tsan", json.dumps(tsan[-1]), flush=True)
summary["tsan"] = tsan
os.makedirs(os.path.dirname(out_json), exist_ok=True)
json.dump(summary, open(out_json, "w"), indent=1)
print("SOAK: %d runs, %d differing from the reference, %d verify lines, %d verify hash mismatches between runs" % (len(summary["runs"]), summary["differing_runs"], summary["verify_lines"], summary["verify_hash_mismatch"]))
shutil.rmtree(d, ignore_errors=True)
