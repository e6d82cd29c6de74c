#!/usr/bin/env python3
"""bench.py -- learned-index seeding throughput of the MI355X backend (BASELINE.json metric).

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one pass of the seeding hot path (P-RMI lookup -> SA last-mile search -> SMEM enumeration,
all three rounds, hit gather included) over one batch of synthetic 150-bp reads that is already resident
in HBM.  Reads shard across ranks with no data-path collective (weak scaling: every GPU gets its own
batch); the index is built once by rank 0 and broadcast over RCCL.  One JSON line is printed by rank 0.

Default workload = BASELINE.json configs[1]: a GRCh38-sized (3.1 Gbp) synthetic genome -- 6.2 G suffixes, 99 GB of
suffix-array entries in HBM -- and 10 M synthetic 150-bp single-end reads per GPU per step, seeding only.  The index
-- suffix array, entries, P-RMI -- is built on the GPU in a few seconds (MEME_BENCH_SA=host: our host builders, minutes,
cached in /dev/shm for the next invocation).

Besides the headline line's `roofline` and `cpu_baseline` objects, rank 0 at N=1 adds `chain` (chaining of 2 M of the batch's reads on the device), `ext` (+ `cigar`), `kswv` (mate-rescue Smith-Waterman jobs), `bsw` (the banded-SW kernel on 2 M
distinct extension jobs) and `e2e` (BASELINE.json's second metric: paired-end `mem -7` through the reference aligner
with the HIP backend bound in, next to the unmodified reference on the same host cores, SAM md5 compared).

Workload knobs (environment): MEME_BENCH_MBP (genome size in Mbp, default 3100), MEME_BENCH_READS (reads per GPU
per step, default 10,000,000), MEME_BENCH_BITS (P-RMI leaves = 2^bits, default: the reference's rule),
MEME_BENCH_LANES (lanes per read in the SA-search kernel), MEME_BENCH_CPU (reference | port | 0; default: the
compiled reference, timed in this run), MEME_BENCH_CPU_READS (sample size), MEME_BENCH_CACHE (0 disables the /dev/shm caches),
MEME_BENCH_SA (device | host: where the suffix array is built), MEME_BENCH_CHAIN / MEME_BENCH_BSW / MEME_BENCH_E2E (0 disables the leg), MEME_BENCH_E2E_PAIRS (read pairs of the end-to-end leg, default
10,000,000 = BASELINE configs[2]), MEME_BENCH_C4 / MEME_BENCH_C4_E2E (0 disables the configs[4]-class leg / its end-to-end part), MEME_BENCH_BUDGET_S (wall budget in seconds after which optional legs are skipped, default 1500).
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
# The chaining stage runs four kernels side by side; the HIP runtime gives a process 4 hardware queues per device unless told otherwise
# (libmeme_hip.so asks for 8 when it is loaded -- too late in a process where torch initialises the runtime first, as here).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pymeme import hipapi, hostapi, synth, workload  # noqa: E402

READ_LEN = 150
BSW_VALU_PER_CELL = 24.3     # measured: profiles/r03_bsw.md (SQ_INSTS_VALU x 64 lanes / DP cells of 2 M distinct pairs, a committed PMC pass)
T_START = time.time()


def host_cpu_quota():
    """CPUs' worth of time the cgroup gives this process (cpu.max = "<quota> <period>" in cgroup v2; cfs_quota_us / cfs_period_us in v1); None: unlimited or unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def log(msg):
    print("[bench r%s +%.0fs] %s" % (os.environ.get("RANK", "0"), time.time() - T_START, msg), file=sys.stderr, flush=True)


def fastq_bytes(reads: np.ndarray) -> np.ndarray:
    """Vectorised FASTQ image of fixed-length reads (constant name/quality; the harness ignores both)."""
    n, L = reads.shape
    row = np.empty((n, 3 + L + 3 + L + 1), dtype=np.uint8)
    row[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
    row[:, 3:3 + L] = np.frombuffer(b"ACGTN", dtype=np.uint8)[reads]
    row[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    row[:, 6 + L:6 + 2 * L] = ord("I")
    row[:, 6 + 2 * L] = ord("\n")
    return row


def reference_index_on_disk(fwd, text, sa, l1, l2, l_pac, bits, keep_others=False):
    """The benchmark index in the reference's own file formats (what `bwa-meme index` + the trainer write), kept next to
    the suffix-array cache in /dev/shm: the compiled reference binaries (cpu_baseline and e2e legs) load it from there."""
    root = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    d = os.path.join(root, "meme_bench_ref_%d_b%d" % (l_pac, bits))
    prefix = os.path.join(d, "ref.fa")
    if os.path.exists(os.path.join(d, "ok")):
        return prefix
    need = 7.3 * 2 * l_pac + 24.0 * l2.shape[0]
    if shutil.disk_usage(root).free < 1.15 * need:
        raise RuntimeError("not enough room in %s for the reference-format index (%.0f GB)" % (root, need / 1e9))
    import glob
    for old in glob.glob(os.path.join(root, "meme_bench_ref_*")):
        if old != d and not keep_others:
            shutil.rmtree(old, ignore_errors=True)
    os.makedirs(d, exist_ok=True)
    t0 = time.time()
    hostapi.write_index(prefix, fwd, text, sa, l1, l2, n_contigs=4)
    open(os.path.join(d, "ok"), "w").write("ok")
    log("reference-format index written to %s in %.1f s" % (d, time.time() - t0))
    return prefix


def cpu_baseline_reference(prefix, reads, cores):
    """Times the COMPILED REFERENCE (oracle/_ref/learned_seeding_mode3 = test/Learned_seeding_big_read.cpp,
    MODE=3, AVX-512 build) on the same index and a sample of the same reads.  Measurement only."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_py as O
    exe = os.path.join(REPO, "oracle", "_ref", "learned_seeding_mode3")
    tmp = tempfile.mkdtemp(prefix="meme_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        fq = os.path.join(tmp, "sample.fq")
        fastq_bytes(reads).tofile(fq)
        hz = O.tsc_hz()
        t0 = time.time()
        env = dict(os.environ, OMP_NUM_THREADS=str(cores))
        r = subprocess.run([exe, prefix, fq, "1000", str(cores), "3"], capture_output=True, text=True, env=env,
                           timeout=3000)
        wall = time.time() - t0
        cyc = None
        for line in r.stderr.splitlines():
            if line.startswith("Consumed:"):
                cyc = float(line.split()[1])
        if r.returncode != 0 or cyc is None:
            raise RuntimeError("reference harness failed: " + r.stderr[-500:])
        secs = cyc / hz
        log("cpu_baseline: reference seeded %d reads on %d threads in %.2f s (process wall %.1f s incl. index load)"
            % (reads.shape[0], cores, secs, wall))
        return {"value": reads.shape[0] / secs, "unit": "reads/s", "cores": cores, "kind": "reference",
                "sample": "%d of the benchmark's reads, same index; seeding region of test/Learned_seeding_big_read.cpp "
                          "(MODE=3, AVX-512 build, steps=3) timed by its own rdtsc counter in this run" % reads.shape[0],
                "process_wall_s": wall}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline_port(text, sa, l1, l2, reads, cores):
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_py as O
    idx = O.Index(text, sa)
    off = np.arange(0, (reads.shape[0] + 1) * READ_LEN, READ_LEN, dtype=np.int64)
    t0 = time.time()
    O.refpath_seed_batch(idx, l1, l2, reads, off, threads=cores, keep_smems=False)
    secs = time.time() - t0
    return {"value": reads.shape[0] / secs, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "%d of the benchmark's reads; oracle/meme_refpath.c (restated MODE-2 probe sequence, OpenMP)"
                      % reads.shape[0]}


def _canon_seeds(rid, start, end, hitbeg, hitcount, hits, hit_base):
    """SMEMs in the reference harness's dump order (per read by start ascending, end descending) with their hit lists flattened in
    that order.  hit_base[i] = index in `hits` of SMEM i's read's first hit."""
    order = np.lexsort((-end.astype(np.int64), start.astype(np.int64), rid))
    rid, start, end, hitbeg, hitcount, hit_base = rid[order], start[order], end[order], hitbeg[order], hitcount[order], hit_base[order]
    tot = int(hitcount.sum())
    first = np.zeros(hitcount.shape[0] + 1, np.int64)
    first[1:] = np.cumsum(hitcount)
    src = np.repeat(hit_base + hitbeg - first[:-1], hitcount) + np.arange(tot, dtype=np.int64)
    return rid, start, end, hitcount, hits[src]


def seeds_equal_oracle(ctx, O, o_idx, part, opt):
    """The GPU's seeds of `part` (through the C ABI) against oracle/meme_oracle.c orc_seed_batch: every SMEM, every hit position."""
    n, rl = part.shape
    off = np.arange(0, (n + 1) * rl, rl, dtype=np.int64)
    g_sm, g_so, g_h, g_ho = ctx.seed_batch(part, off, opt)
    cap, hcap = 256, 2048
    while True:
        try:
            o_sm, o_ns, o_h, o_nh, _ = O.seed_batch(o_idx, part, off, smem_cap=cap, hit_cap=hcap, threads=0)
            break
        except RuntimeError:                                  # a read with more SMEMs / hits than the slice's arrays hold
            cap, hcap = cap * 4, hcap * 8
    if not np.array_equal(np.diff(g_so), o_ns.astype(np.int64)):
        return False
    g_rid = np.repeat(np.arange(n, dtype=np.int64), np.diff(g_so))
    a = _canon_seeds(g_rid, g_sm["start"], g_sm["end"], g_sm["hitbeg"].astype(np.int64), g_sm["hitcount"].astype(np.int64), g_h, g_ho[:-1][g_rid])
    mask = np.arange(cap)[None, :] < o_ns[:, None]
    o_flat = o_sm[mask]
    b = _canon_seeds(g_rid, o_flat["start"], o_flat["end"], o_flat["hitbeg"].astype(np.int64), o_flat["hitcount"].astype(np.int64), o_h.reshape(-1),
                     (g_rid * hcap))
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def algorithmic_bytes_per_read(text, sa, l1, l2, reads):
    """SURVEY 8(d) per-read figure from the instrumented restatement, on a sample of the same reads."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_py as O
    idx = O.Index(text, sa)
    n, rl = reads.shape
    off = np.arange(0, (n + 1) * rl, rl, dtype=np.int64)
    _, _, ctr = O.refpath_seed_batch(idx, l1, l2, reads, off, threads=0, keep_smems=False)
    return O.algorithmic_bytes(ctr, n * rl) / n, {k: v / n for k, v in ctr.items()}


def leg_contigs(l_pac, k=4):
    """The benchmark genome as k reference sequences (a bntann1_t length is 32-bit: one 3.1 Gbp contig is not a valid reference)."""
    cut = [l_pac * i // k for i in range(k + 1)]
    return [(cut[i], cut[i + 1] - cut[i], 0) for i in range(k)]


def chain_leg(ctx, reads, l_pac, nsub=2000000):
    """Chaining on the device (mem_chain_Learned + mem_chain_flt): the first `nsub` reads of the benchmark batch
    are seeded through the pinned-result call and chained where their seeds lie; kernel time by HIP events, parity of `ncheck` reads
    against the oracle's restatement (pinned on the compiled reference's chains, tests/golden/chain_golden.npz)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_py
    n, rl = min(nsub, reads.shape[0]), reads.shape[1]
    off = np.arange(0, (n + 1) * rl, rl, dtype=np.int64)
    smems, smem_off, hits, hit_off = ctx.seed_batch_host(reads[:n].reshape(-1), off)
    contigs = leg_contigs(l_pac)
    res = ctx.chain_last_batch_host(contigs, hipapi.default_chain_opt(l_pac))
    res = ctx.chain_last_batch_host(contigs, hipapi.default_chain_opt(l_pac))          # (second call: buffers exist)
    tm = ctx.timings()
    # every read of the leg against the oracle's restatement (batched C call, all host cores)
    copt = oracle_py.default_chain_opt(l_pac)
    n_bad, first_bad = oracle_py.chain_compare_batch(smems, smem_off, hits, hit_off, np.full(n, rl, np.int32), np.array([c[0] for c in contigs], np.int64),
                                                     np.zeros(len(contigs), np.uint8), copt, res)
    same, checked = n_bad == 0 and res["n_fallback"] == 0, n
    if not same:
        log("chain leg: %d reads differ from the oracle (first: %d), %d left to the host" % (n_bad, first_bad, res["n_fallback"]))
    return {"metric": "chain_reads_per_sec", "value": n / (tm.chain_kernel_ms * 1e-3) if same and tm.chain_kernel_ms > 0 else None, "unit": "reads/s",
            "reads": n, "kernel_ms": tm.chain_kernel_ms, "wavefront_tiers_ms": tm.chain_pass2_ms, "btree_tier_ms": tm.chain_tier3_ms, "btree_tier_reads": int(tm.chain_tier3_reads), "chains": int(res["chains"].shape[0]),
            "chained_seeds": int(res["seeds"].shape[0]), "reads_left_to_host": int(res["n_fallback"]), "reads_in_wavefront_tier": int(res["n_tier2"]),
            "matches_oracle": bool(same),
            "checked_reads": checked}


def infer_bw(l1, l2, score, a, q, r):
    """infer_bw (src/bwamem.cpp:2151-2158), vectorised"""
    w = ((np.minimum(l1, l2) * a - score - q) / float(r) + 2.).astype(np.int64)          # (the C cast truncates toward zero, as astype does)
    w = np.maximum(w, np.abs(l1 - l2))
    return np.where((l1 == l2) & (l1 * a - score < ((q + r - a) << 1)), 0, w)


def ext_leg(ctx, reads, genome, l_pac, nsub=2000000, ncig=400000, check_reads=None):
    """The stages behind seeding, on the device: chaining + seed extension of `nsub` of the reads without leaving
    HBM (meme_extend_last_batch_host: the host receives alignment records), then bwa_gen_cigar2 whole (meme_gen_cigar_batch_host: CIGAR, NM,
    MD) for the best record of `ncig` reads, called with the band argument mem_reg2aln would pass first.  Kernel times by HIP events; EVERY
    record is compared with the oracle's restatement of mem_chain2aln_across_reads_V2 (orc_extend_batch on the device's chains, themselves
    checked by the chain leg), and CIGAR / NM / MD of a sample with orc_gen_cigar2 (pinned on the compiled reference's function)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_py
    n, rl = min(nsub, reads.shape[0]), reads.shape[1]
    off = np.arange(0, (n + 1) * rl, rl, dtype=np.int64)
    ctx.seed_batch_host(reads[:n].reshape(-1), off)
    contigs = leg_contigs(l_pac)
    copt = hipapi.default_chain_opt(l_pac)
    R = ctx.extend_last_batch_host(contigs, copt)
    R = ctx.extend_last_batch_host(contigs, copt)          # (second call: buffers exist)
    ctx.set_tuning("ext_census", 1)                        # (a third, untimed call: what the jobs are -- LDS classes, band cells, how many a closed form could answer)
    C3 = ctx.extend_last_batch_host(contigs, copt)
    ctx.set_tuning("ext_census", 0)
    ch = ctx.chain_last_batch_host(contigs, copt)
    text = hipapi.fwd_rc_text(genome)
    # MEME_BENCH_EXT_CHECK=0 is for kernel probes only (the oracle's pass over 2 M reads takes minutes of host time): the leg then reports NO throughput value
    # and matches_oracle null -- a number that was not checked is not a measurement -- while the CIGAR sub-leg keeps its own sampled check.
    checked = os.environ.get("MEME_BENCH_EXT_CHECK", "1") != "0"
    # (check_reads: the oracle's pass over the first so many reads of the leg -- reads are extended independently of each other, the arrays are indexed by read;
    # the 250-bp class poses 15 jobs per read and its full pass was 90 s of the default run's wall -- the timed calls above are over all n reads either way)
    m = n if check_reads is None else min(n, int(check_reads))
    if checked:
        want, (jobs, retried) = oracle_py.extend_batch(reads[:m].reshape(-1), off[:m + 1], ch["chain_off"][:m + 1], oracle_py.chains_as_orc(ch["chains"]), ch["seed_off"][:m + 1],
                                                      ch["seeds"], ch["frac_rep"][:m], text, l_pac, np.array([c[0] for c in contigs], np.int64),
                                                      np.array([c[1] for c in contigs], np.int32))
        nrec = int(ch["seed_off"][m])
        same = np.array_equal(R["reg_off"], ch["seed_off"]) and all(np.array_equal(R["regs"][f][:nrec].astype(np.int64), want[f].astype(np.int64)) for f in oracle_py.ALNREG_FIELDS)
    else:
        want, same = np.zeros(0, dtype=R["regs"].dtype), False
    first = max(R["n_pairs"] - R["n_retried"], 1)
    cls = C3["census_class"]
    out = {"metric": "extend_reads_per_sec", "value": n / ((R["chain_ms"] + R["ext_ms"]) * 1e-3) if same else None, "unit": "reads/s", "reads": n, "read_len": rl,
           "chain_ms": R["chain_ms"], "ext_ms": R["ext_ms"], "bsw_ms": R["bsw_ms"], "alignment_records": int(R["regs"].shape[0]), "extension_jobs": R["n_pairs"],
           "jobs_with_doubled_band": R["n_retried"], "reads_in_wavefront_tiers": R["n_tier2"], "matches_oracle": bool(same) if checked else None, "checked_records": int(want.shape[0]), "checked_reads": int(m) if checked else 0,
           "exact_prefix_jobs": int(C3["n_exact_prefix"]), "exact_prefix_share": C3["n_exact_prefix"] / first,
           # cells of the jobs' band-limited matrices (first attempts; the kernel trims rows and stops at z-drop, so it evaluates fewer)
           "band_cells_per_job": C3["census_band_cells"] / first, "gcups_band_cells": (C3["census_band_cells"] / 1e9) / (R["bsw_ms"] * 1e-3) if R["bsw_ms"] > 0 else None,
           "jobs_by_lds_class": {"query<=%d" % q: cls[i] / first for i, q in enumerate((30, 62, 94, 126, 158, 222, 318, 600))} | {"query>600": cls[8] / first}}
    # the stage as the bound aligner runs it (tuning "ext_live_only": extension in rounds, surviving records only): must be the records above minus the
    # purged ones -- which were just compared with the oracle one by one
    ctx.set_tuning("ext_live_only", 1)
    RL = ctx.extend_last_batch_host(contigs, copt)
    RL = ctx.extend_last_batch_host(contigs, copt)
    sweep = {}
    for t in [int(x) for x in os.environ.get("MEME_BENCH_EXT_ROUNDS", "").split(",") if x]:        # (probe: other numbers of one-seed rounds)
        ctx.set_tuning("ext_rounds", t)
        ctx.extend_last_batch_host(contigs, copt)
        X = ctx.extend_last_batch_host(contigs, copt)
        sweep[str(t)] = {"ext_ms": X["ext_ms"], "bsw_ms": X["bsw_ms"], "extension_jobs": X["n_pairs"], "seeds_extended": X["n_ext_seeds"], "bsw_launches": X["n_bsw_calls"],
                         "same_records": hipapi.records_equal(X["regs"], RL["regs"])}
    if sweep:
        ctx.set_tuning("ext_rounds", 1)
    ctx.set_tuning("ext_live_only", 0)
    keep = R["regs"]["qe"] > R["regs"]["qb"]
    same_l = same and np.array_equal(RL["reg_off"], np.concatenate([[0], np.cumsum(keep.astype(np.int64))])[R["reg_off"]]) and hipapi.records_equal(RL["regs"], R["regs"][keep])
    out["all_seeds_at_once"] = {"value": out["value"], "ext_ms": R["ext_ms"], "bsw_ms": R["bsw_ms"], "extension_jobs": R["n_pairs"]}
    out["in_rounds"] = {"value": n / ((RL["chain_ms"] + RL["ext_ms"]) * 1e-3) if same_l else None, "chain_ms": RL["chain_ms"], "ext_ms": RL["ext_ms"], "bsw_ms": RL["bsw_ms"],
                        "extension_jobs": RL["n_pairs"], "jobs_with_doubled_band": RL["n_retried"], "bsw_launches": RL["n_bsw_calls"], "chained_seeds": RL["total_seeds"],
                        "seeds_extended": RL["n_ext_seeds"], "records_handed_to_the_host": int(RL["regs"].shape[0]),
                        "equals_the_checked_records_minus_the_purged_ones": bool(same_l)} | ({"other_round_counts": sweep} if sweep else {})
    if not checked:
        out["check"] = "SKIPPED (MEME_BENCH_EXT_CHECK=0, a kernel probe): no record of this leg was compared with the oracle, so it reports no value"
        out["in_rounds"]["equals_the_checked_records_minus_the_purged_ones"] = None
    if same_l:
        out["value"] = out["in_rounds"]["value"]
        out["what"] = "chaining + extension in rounds (surviving records only), as the bound aligner calls the stage; all_seeds_at_once = the reference's batch order, every record checked"
    # bwa_gen_cigar2 for every read's best live record, called as mem_reg2aln calls it first (src/bwamem.cpp:2333-2342 with a 1, o 6, e 1, w 100)
    regs, ro = R["regs"], R["reg_off"]
    rid = np.repeat(np.arange(n), np.diff(ro))
    live = np.nonzero((regs["qe"] > regs["qb"]) & (regs["rb"] >= 0) & (regs["re"] > regs["rb"]) & (rid < ncig) & ((regs["rb"] < l_pac) == (regs["re"] <= l_pac)))[0]
    live = live[np.lexsort((-regs["score"][live].astype(np.int64), rid[live]))]
    best = live[np.concatenate([[True], rid[live][1:] != rid[live][:-1]])] if live.shape[0] else live
    ql = (regs["qe"][best] - regs["qb"][best]).astype(np.int64)
    tl = (regs["re"][best] - regs["rb"][best]).astype(np.int64)
    sc = regs["truesc"][best].astype(np.int64)
    w2 = np.maximum(infer_bw(ql, tl, sc, 1, 6, 1), infer_bw(ql, tl, sc, 1, 6, 1))
    w2 = np.where(w2 > 100, np.minimum(w2, regs["w"][best].astype(np.int64)), w2)
    w2 = np.minimum(w2, 400)
    J = np.zeros(best.shape[0], dtype=hipapi.CJOB)
    J["rb"], J["read"], J["qb"], J["qlen"], J["tlen"], J["w_"] = regs["rb"][best], rid[best], regs["qb"][best], ql, tl, w2
    res, cig, md, ms = ctx.gen_cigar_batch_host(J)
    res, cig, md, ms = ctx.gen_cigar_batch_host(J)
    ok = True
    step = max(1, J.shape[0] // 20000)
    for k in range(0, J.shape[0], step):
        j = J[k]
        q = reads[int(j["read"])][int(j["qb"]):int(j["qb"]) + int(j["qlen"])]
        o_sc, o_cg, o_nm, o_md = oracle_py.gen_cigar2(text, l_pac, q, int(j["rb"]), int(j["rb"]) + int(j["tlen"]), int(j["w_"]))
        r = res[k]
        o0, m0 = int(r["cigar_off"]), int(r["md_off"])
        ok = ok and o_sc == int(r["score"]) and o_nm == int(r["nm"]) and np.array_equal(o_cg, cig[o0:o0 + int(r["n_cigar"])]) and o_md == md[m0:m0 + int(r["md_len"])].tobytes()
    nogap = int(np.sum((ql == tl) & (w2 == 0)))
    cls = list(ctx.timings().gcig_class_jobs)
    out["cigar"] = {"jobs_by_kernel": {"k_gcig_grp<16> (4 jobs per wavefront)": int(cls[0]), "k_gcig_grp<32> (2 per wavefront)": int(cls[1]), "k_gcig_grp<64> (one per wavefront, one chunk per row)": int(cls[4]), "k_gcig_t<true> (a wavefront each, two columns per lane)": int(cls[5]), "k_gcig_t<false> (a wavefront each, 64-column chunks)": int(cls[2]),
                                       "k_gcig_nogap (a lane each)": int(cls[3])}}
    out["cigar"] |= {"metric": "cigar_alignments_per_sec", "value": J.shape[0] / (ms * 1e-3) if ok and ms > 0 else None, "unit": "alignments/s", "alignments": int(J.shape[0]),
                    "what": "bwa_gen_cigar2 whole (meme_gen_cigar_batch_host): CIGAR + NM + MD", "gap_free_shortcut_share": nogap / max(J.shape[0], 1),
                    "kernel_ms": ms, "operations": int(cig.shape[0]), "md_bytes": int(md.shape[0]), "matches_oracle": bool(ok), "checked": int(len(range(0, J.shape[0], step)))}
    return out


def bsw_reference_baseline(pairs, ref, qer, w, cores):
    """The reference's own AVX-512 kernels (BandedPairWiseSW::getScores8 / getScores16 behind oracle/_ref/libbsw_ref.so) on all host
    cores: pairs classed the way mem_chain2aln_across_reads_V2 classes them (src/bwamem.cpp:2452: both lengths < 128 and
    h0 + min(len) * a < 128 -> int8 lanes, else int16), each class sorted by query length like sortPairsLenExt so the SIMD lanes fill,
    512-pair calls spread over a thread pool (ctypes releases the GIL).  None when the compiled reference is not there."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    import oracle_py, ref_py
    if not (ref_py.have("libbsw_ref.so") and ref_py.cpu_can_run()):
        return None
    L = ref_py.bsw_lib()
    par = oracle_py.default_bsw_params()
    refp = np.concatenate([np.ascontiguousarray(ref, dtype=np.uint8), np.zeros(1 << 16, np.uint8)])
    qerp = np.concatenate([np.ascontiguousarray(qer, dtype=np.uint8), np.zeros(1 << 16, np.uint8)])
    P = pairs.view(oracle_py.SEQPAIR_DTYPE)
    small = (P["len1"] < 128) & (P["len2"] < 128) & (P["h0"] + np.minimum(P["len1"], P["len2"]) * par.a < 128)
    jobs = []
    for kind, idx in ((8, np.nonzero(small)[0]), (16, np.nonzero(~small)[0])):
        idx = idx[np.argsort(P["len2"][idx], kind="stable")]
        for b in range(0, idx.shape[0], 512):
            sub = np.zeros(min(512, idx.shape[0] - b) + 128, dtype=oracle_py.SEQPAIR_DTYPE)    # (the SIMD wrappers pad past numPairs)
            sub[:-128] = P[idx[b:b + 512]]
            jobs.append((kind, sub, idx[b:b + 512]))
    def run(j):
        kind, sub, _ = j
        return L.ref_bsw_run(C.c_int(kind), C.c_void_p(sub.ctypes.data), C.c_void_p(refp.ctypes.data), C.c_void_p(qerp.ctypes.data),
                             C.c_int32(sub.shape[0] - 128), C.c_int32(w), C.byref(par))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        rcs = list(ex.map(run, jobs))
    dt = time.perf_counter() - t0
    assert all(r == 0 for r in rcs)
    out = P.copy()
    for kind, sub, idx in jobs:
        out[idx] = sub[:-128]
    return dt, out, int(small.sum())


def kswv_leg(ctx):
    """Mate-rescue Smith-Waterman of the SAM phase (mem_sam_pe_batch / kswv): jobs shaped like those of 150-bp pairs (a mate
    against a window of a few hundred bases around where its partner says it should lie), results to the host as the binding takes them.
    jobs/s through the C ABI call (transfers included), the kernel's own time, GCUPS with the DP cells from the oracle's counter; checked
    against the oracle; CPU figure from the reference's own batch (sort_classify + mem_sam_pe_batch, AVX-512 kswv kernels) on all cores in
    batches of the size worker_sam sees, the oracle's scalar restatement beside it."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    import oracle_py, ref_py
    from common import kswv_workload
    njobs = int(os.environ.get("MEME_BENCH_KSWV_JOBS", "200000"))
    base, ref, qer = kswv_workload(n=20000, seed=123, read_len=(150, 151))
    reps = (njobs + base.shape[0] - 1) // base.shape[0]
    jobs = np.concatenate([base] * reps)[:njobs].copy()                    # (the same bytes serve several jobs: the device does not modify them)
    ms, walls = [], []
    for it in range(3):
        t0 = time.perf_counter()
        got, k_ms = ctx.kswv_batch_host(jobs.view(hipapi.KSWV_JOB), ref, qer)
        walls.append(time.perf_counter() - t0)
        ms.append(k_ms)
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    want, cells = oracle_py.kswv_batch(base, ref, qer, threads=cores)
    port_dt = time.perf_counter() - t0
    same = bool(np.array_equal(got[:base.shape[0]].view(np.int32), want.view(np.int32)) and np.array_equal(got[-base.shape[0]:].view(np.int32), got[:base.shape[0]].view(np.int32))) \
        if njobs % base.shape[0] == 0 else bool(np.array_equal(got[:base.shape[0]].view(np.int32), want.view(np.int32)))
    cells_per_job = cells / base.shape[0]
    k = float(np.min(ms))
    cpu = {"value": base.shape[0] / port_dt, "unit": "jobs/s", "cores": cores, "kind": "port",
           "sample": "%d jobs, scalar restatement (orc_kswv_batch) on %d threads" % (base.shape[0], cores)}
    if ref_py.have("libstage_ref.so") and ref_py.cpu_can_run() and ref_py.stage_lib().ref_kswv_batch(None, 0, None, 0, None, 0, 1, 4, 6, 1, 6, 1, None) == 0:
        per = 64                                                            # jobs per call: what one worker batch of 256 pairs poses, give or take
        parts = []
        for i in range(0, base.shape[0], per):                              # (inputs prepared outside the timed region: the calls themselves release the GIL)
            j = base[i:i + per].copy()
            r0, q0 = int(j["idr"][0]), int(j["idq"][0])
            r1, q1 = int(j["idr"][-1] + j["len1"][-1]), int(j["idq"][-1] + j["len2"][-1])
            j["idr"] -= r0; j["idq"] -= q0
            parts.append((j, ref[r0:r1].copy(), qer[q0:q1].copy()))
        def run(p):
            return ref_py.kswv_batch(p[0], p[1], p[2])
        with ThreadPoolExecutor(max_workers=cores) as ex:
            list(ex.map(run, parts[:cores]))                                # (threads started, library loaded)
            t0 = time.perf_counter()
            outs = list(ex.map(run, parts))
        r_dt = time.perf_counter() - t0
        r_same = bool(np.array_equal(np.concatenate(outs).view(np.int32), got[:base.shape[0]].view(np.int32)))
        cpu = {"value": base.shape[0] / r_dt, "unit": "jobs/s", "cores": cores, "kind": "reference",
               "sample": "%d jobs: sort_classify + mem_sam_pe_batch (AVX-512 kswv kernels) in calls of %d jobs on %d threads" % (base.shape[0], per, cores),
               "device_matches_reference": r_same, "port": cpu}
    return {"metric": "kswv_jobs_per_sec", "value": njobs / float(np.min(walls)), "unit": "jobs/s", "jobs": njobs, "call_ms": float(np.min(walls)) * 1e3, "kernel_ms": k,
            "cells_per_job": cells_per_job, "gcups_kernel": cells_per_job * njobs / (k * 1e-3) / 1e9, "matches_oracle": same, "checked_jobs": int(base.shape[0]),
            "cpu_baseline": cpu}


def bsw_leg(ctx, dev, world):
    """Second kernel of the path: banded seed extension on rank 0's GPU, pairs resident in HBM.
    Every pair is distinct (no tiling): lengths, targets and errors differ from pair to pair as in a real batch, so the
    lanes of a wavefront diverge the way they do in production.  Reported beside the headline metric: pairs/s, GCUPS
    (DP cells from the CPU restatement's counter on a sample) and the CPU restatement on all host cores."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_py
    npairs = int(os.environ.get("MEME_BENCH_BSW_PAIRS", "2000000"))
    pairs, ref, qer = workload.make_bsw_pairs_distinct(npairs, seed=77, read_len=READ_LEN)
    d_pairs = torch.from_numpy(pairs.view(np.uint8)).to(dev)
    d_ref = torch.from_numpy(ref).to(dev)
    d_qer = torch.from_numpy(qer).to(dev)
    torch.cuda.synchronize()
    ms = []
    for it in range(4):
        ctx.bsw_batch_device(d_pairs.data_ptr(), d_ref.data_ptr(), d_qer.data_ptr(), npairs, 100)
        ctx.sync()
        if it:
            ms.append(ctx.timings().bsw_kernel_ms)
    k_ms = float(np.mean(ms))
    # cells and the CPU figure from the restatement of scalarBandedSWA (the oracle): checker + baseline only
    ns = min(npairs, 400000)
    got = d_pairs.cpu().numpy().view(hipapi.SEQPAIR)[:ns]
    chk = pairs[:ns].copy().view(oracle_py.SEQPAIR_DTYPE)
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    cells = oracle_py.bsw_batch(chk, ref, qer, 100, threads=cores)
    cpu_dt = time.perf_counter() - t0
    same = all(np.array_equal(got[f], chk[f]) for f in ("score", "tle", "gtle", "qle", "gscore", "max_off"))
    cells_per_pair = cells / ns
    gcups = cells_per_pair * npairs / (k_ms * 1e-3) / 1e9
    cpu = {"value": ns / cpu_dt, "unit": "pairs/s", "cores": cores, "kind": "port",
           "sample": "%d pairs, scalar restatement of scalarBandedSWA on %d threads" % (ns, cores)}
    rb = bsw_reference_baseline(pairs[:ns].copy(), ref, qer, 100, cores)
    if rb is not None:
        r_dt, r_out, n8 = rb
        r_same = all(np.array_equal(got[f], r_out[f]) for f in ("score", "tle", "gtle", "qle", "gscore", "max_off"))
        cpu = {"value": ns / r_dt, "unit": "pairs/s", "cores": cores, "kind": "reference",
               "sample": "%d pairs: BandedPairWiseSW::getScores8 (%d pairs) / getScores16 (%d pairs) of the AVX-512 build, classed and length-sorted as in "
                         "mem_chain2aln_across_reads_V2, 512-pair calls on %d threads" % (ns, n8, ns - n8, cores),
               "device_matches_reference": bool(r_same), "port": cpu}
    return {"metric": "bsw_pairs_per_sec", "value": npairs / (k_ms * 1e-3), "unit": "pairs/s", "per": "gpu",
            "pairs": npairs, "distinct_pairs": npairs, "band_w": 100, "kernel_ms": k_ms, "cells_per_pair": cells_per_pair,
            "gcups": gcups, "matches_oracle": bool(same), "checked_pairs": ns,
            # integer-VALU roofline of the lane-per-pair kernel: VALU instructions per DP cell from the committed counter pass
            # (profiles/r03_bsw.md, SQ_INSTS_VALU / cells), against 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz int32 lane-ops/s
            "roofline": {"bound": "valu-int32", "peak": 39.3, "unit": "Tops/s", "ops_per_cell": BSW_VALU_PER_CELL,
                         "ops_per_cell_source": "SQ_INSTS_VALU x 64 lanes / DP cells, profiles/r03_bsw.md",
                         "achieved": BSW_VALU_PER_CELL * gcups / 1e3, "frac": BSW_VALU_PER_CELL * gcups / 1e3 / 39.3},
            "cpu_baseline": cpu}


def config4_class_leg(ctx, dev, genome, text, sa, l1, l2, l_pac, steps=3):
    """BASELINE configs[4]'s read class on the benchmark index, one GPU: 250-bp reads with 5 % substitutions and 0.75 % single-base indels
    (both mates of synthetic pairs, so both strands).  Seeding reads/s with its own algorithmic bytes per read and roofline fraction, then
    the stages behind it (chaining, extension with the banded-SW jobs' LDS classes and band cells, bwa_gen_cigar2 whole), every stage
    checked against the oracle as the 150-bp legs are.  The end-to-end part of the class is run by main() after the HBM is free."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_py as O
    rl, sub, indel = 250, 0.05, 0.0075
    nreads = int(os.environ.get("MEME_BENCH_C4_READS", "2000000"))
    rng = np.random.default_rng(4004)
    parts = []
    for p0 in range(0, nreads // 2, 1 << 18):
        r1, r2 = workload.make_pairs_chunk(genome, min(1 << 18, nreads // 2 - p0), rl, rng, sub, indel)
        parts += [r1, r2]
    reads = np.ascontiguousarray(np.concatenate(parts))
    nreads = reads.shape[0]
    d_reads = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_off = torch.arange(0, (nreads + 1) * rl, rl, dtype=torch.int64, device=dev)
    opt = hipapi.default_seed_opt(rounds=3)
    res = ctx.seed_batch_device(d_reads.data_ptr(), d_off.data_ptr(), nreads, nreads * rl, opt)
    ctx.sync()
    k_ms, wall = [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        res = ctx.seed_batch_device(d_reads.data_ptr(), d_off.data_ptr(), nreads, nreads * rl, opt)
        ctx.sync()
        wall.append(time.perf_counter() - t0)
        tm = ctx.timings()
        k_ms.append((tm.seed_kernel_ms, tm.seed_gather_ms + tm.seed_pack_ms, tm.seed_reseed_ms, tm.seed_windows))
    del d_reads, d_off
    stage_ms = float(np.mean([k[0] for k in k_ms]))
    bpr, per_read = algorithmic_bytes_per_read(text, sa, l1, l2, reads[:8000])
    o_idx = O.Index(text, sa)
    npar = min(nreads, int(os.environ.get("MEME_BENCH_C4_PARITY_READS", "100000")))
    parity = all(seeds_equal_oracle(ctx, O, o_idx, reads[p0:p0 + 25000], opt) for p0 in range(0, npar, 25000))
    achieved = bpr * nreads / (stage_ms * 1e-3) / 1e9
    out = {"workload": "%d reads of %d bp, %g %% substitutions, %g %% single-base indels, both strands, vs the benchmark genome (%d bp); reads resident in HBM"
                       % (nreads, rl, 100 * sub, 100 * indel, l_pac),
           "seeding": {"metric": "seeding_reads_per_sec", "value": nreads / float(np.mean(wall)) if parity else None, "unit": "reads/s", "reads": nreads, "steps": steps,
                       "ms_per_step": float(np.mean(wall)) * 1e3, "search_stage_ms": stage_ms, "of_which_reseed_kernels_ms": float(np.mean([k[2] for k in k_ms])),
                       "pack_gather_ms": float(np.mean([k[1] for k in k_ms])), "smems_per_read": res.total_smems / nreads, "hits_per_read": res.total_hits / nreads,
                       "searches_per_read": res.searches / nreads, "windows_per_read": k_ms[-1][3] / nreads,
                       "algorithmic_bytes_per_read": bpr, "work_per_read": per_read,
                       "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0},
                       "matches_oracle": bool(parity), "checked_reads": npar}}
    out["chain"] = chain_leg(ctx, reads, l_pac, nsub=nreads)
    out["ext"] = ext_leg(ctx, reads, genome, l_pac, nsub=nreads, ncig=min(nreads, 400000), check_reads=int(os.environ.get("MEME_BENCH_C4_EXT_CHECK_READS", "600000")))
    return out


def repeat_dense_leg(device_index, steps=3, want_ref_index=False):
    """Every stage of the path on a REPEAT-DENSE genome: seeding (the overflow tiers, max_occ subsampling of hit lists, src/bwamem.cpp:1154-1160),
    chaining (the wavefront-per-read tiers and the B-tree tier carry load here), extension (jobs per read), each checked against the oracle.
    Own index (built on the device in seconds) on a ctx of its own beside the benchmark's."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_py as O
    mbp = float(os.environ.get("MEME_BENCH_RD_MBP", "128"))
    nreads = int(os.environ.get("MEME_BENCH_RD_READS", "1000000"))
    nsub = int(os.environ.get("MEME_BENCH_RD_STAGE_READS", "200000"))
    l_pac = int(mbp * 1e6) & ~1
    n = 2 * l_pac
    t0 = time.time()
    g = workload.repeat_dense_genome(l_pac)
    c = hipapi.Context(device_index)
    text = hipapi.fwd_rc_text(g)
    d_text, d_sa = hipapi.build_sa_device(c, text)
    d_pos5 = hipapi.pos5_from_sa_torch(c, d_sa, n)
    sa = d_sa.cpu().numpy().view(np.uint64)
    del d_sa
    d_pac, d_ent = hipapi.stage_entries_torch(c, n, d_text, d_pos5)
    bits = 26 if 8.0 * n + 8 > 1.0e9 else 24
    d_l2, n_l2, d_l1, n_l1 = hipapi.train_prmi_device(c, d_ent, n, bits)
    l2 = d_l2.cpu().numpy().view(hostapi.RMI_DTYPE)
    l1 = d_l1.cpu().numpy().view(hostapi.RMI_DTYPE)[:n_l1]
    keep = (d_pac, d_ent) + hipapi.attach_index_torch(c, n, d_pac, d_ent, d_l2, n_l2, d_l1, n_l1)
    del d_text, d_pos5
    log("repeat-dense leg: %.0f Mbp genome + index (2^%d leaves, %d partial) on the device in %.1f s" % (mbp, bits, n_l1, time.time() - t0))
    reads = workload.make_reads_fast(g, nreads, READ_LEN, seed=78)
    dev = torch.device("cuda", device_index)
    d_reads = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_off = torch.arange(0, (nreads + 1) * READ_LEN, READ_LEN, dtype=torch.int64, device=dev)
    opt = hipapi.default_seed_opt(rounds=3)
    res = c.seed_batch_device(d_reads.data_ptr(), d_off.data_ptr(), nreads, nreads * READ_LEN, opt)
    c.sync()
    k_ms, wall = [], []
    for _ in range(steps):
        t1 = time.perf_counter()
        res = c.seed_batch_device(d_reads.data_ptr(), d_off.data_ptr(), nreads, nreads * READ_LEN, opt)
        c.sync()
        wall.append(time.perf_counter() - t1)
        tm = c.timings()
        k_ms.append((tm.seed_kernel_ms, tm.seed_gather_ms + tm.seed_pack_ms, tm.seed_reseed_ms, tm.seed_windows))
    del d_reads, d_off
    stage_ms = float(np.mean([k[0] for k in k_ms]))
    bpr, per_read = algorithmic_bytes_per_read(text, sa, l1, l2, reads[:8000])
    o_idx = O.Index(text, sa)
    npar = min(nreads, int(os.environ.get("MEME_BENCH_RD_PARITY_READS", "40000")))
    parity = all(seeds_equal_oracle(c, O, o_idx, reads[p0:p0 + 10000], opt) for p0 in range(0, npar, 10000))
    achieved = bpr * nreads / (stage_ms * 1e-3) / 1e9
    out = {"workload": "%d reads of %d bp (1 %% substitutions) vs a %d-bp genome with 42 %% interspersed 300-bp repeats (30 %% in 24 families at 12 %% divergence, 12 %% in 2 young families at 2 %%), "
                       "8 satellites (171-bp monomer x 300), 24 exact 3-kb duplications; index of its own, 2^%d leaves" % (nreads, READ_LEN, l_pac, bits),
           "seeding": {"metric": "seeding_reads_per_sec", "value": nreads / float(np.mean(wall)) if parity else None, "unit": "reads/s", "reads": nreads, "steps": steps,
                       "ms_per_step": float(np.mean(wall)) * 1e3, "search_stage_ms": stage_ms, "of_which_reseed_kernels_ms": float(np.mean([k[2] for k in k_ms])),
                       "pack_gather_ms": float(np.mean([k[1] for k in k_ms])), "smems_per_read": res.total_smems / nreads, "hits_per_read": res.total_hits / nreads,
                       "searches_per_read": res.searches / nreads, "windows_per_read": k_ms[-1][3] / nreads, "algorithmic_bytes_per_read": bpr, "work_per_read": per_read,
                       "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0},
                       "matches_oracle": bool(parity), "checked_reads": npar}}
    del o_idx
    out["chain"] = chain_leg(c, reads, l_pac, nsub=nsub)
    out["ext"] = ext_leg(c, reads, g, l_pac, nsub=min(nsub, int(os.environ.get("MEME_BENCH_RD_EXT_READS", "100000"))), ncig=20000)
    e = out["ext"]
    if isinstance(e, dict) and e.get("reads"):
        e["extension_jobs_per_read"] = e["extension_jobs"] / e["reads"]
        e["alignment_records_per_read"] = e["alignment_records"] / e["reads"]
    if want_ref_index:                       # the e2e runs on this genome (main): the reference's own file formats, 2 GB beside the benchmark's
        try:
            out["_ref_prefix"] = reference_index_on_disk(g, text, sa, l1, l2, l_pac, bits, keep_others=True)
            out["_genome"] = g
        except Exception as e2:
            log("repeat-dense leg: no reference-format index (%r)" % (e2,))
    out["all_checks_true"] = bool(parity and out["chain"].get("matches_oracle") and e.get("matches_oracle") and e.get("in_rounds", {}).get("equals_the_checked_records_minus_the_purged_ones") and e.get("cigar", {}).get("matches_oracle"))
    c.close()
    del keep
    torch.cuda.empty_cache()
    return out


def live_pmc_traffic(steps=2, reads_file=None, expect=None):
    try:
        return _live_pmc_traffic(steps, reads_file, expect)
    finally:
        if reads_file and os.path.exists(reads_file):
            os.remove(reads_file)


def _live_pmc_traffic(steps, rfile, expect):
    """HBM traffic of the SA-search stage measured in THIS run: bench.py re-executes itself (seeding only, `steps` launches, no warm-up) once under
    `rocprofv3 --pmc FETCH_SIZE` and once under `--pmc WRITE_SIZE` -- counters in passes of their own, as the MI355X guide prescribes -- and sums the
    counters over the stage's kernels (k_seed<G> + k_reseed*).  Returns {"fetch_kb", "write_kb"} per launch, or None when the tool is not there or a
    pass fails (the caller then keeps the committed pass)."""
    import glob
    import sqlite3
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    out = {}
    for key, counter in (("fetch_kb", "FETCH_SIZE"), ("write_kb", "WRITE_SIZE")):
        d = tempfile.mkdtemp(prefix="meme_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp", **({"MEME_BENCH_READS_FILE": rfile} if rfile else {}), MEME_BENCH_PMC="0", MEME_BENCH_CPU="0", MEME_BENCH_E2E="0", MEME_BENCH_BSW="0", MEME_BENCH_KSWV="0", MEME_BENCH_CHAIN="0",
                       MEME_BENCH_EXT="0", MEME_BENCH_C4="0", MEME_BENCH_RD="0", MEME_BENCH_PARITY_READS="0", MEME_BENCH_NO_FALLBACK="1")
            r = subprocess.run([exe, "--pmc", counter, "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", "0"],
                               cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            dbs = glob.glob(os.path.join(d, "**", "*results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                log("pmc pass %s failed (rc %d): %s" % (counter, r.returncode, r.stderr.decode(errors="replace")[-300:]))
                return None
            cur = sqlite3.connect(dbs[0]).cursor()
            rows = cur.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
            stage = [x for x in rows if ("k_seed" in x[0] or "k_reseed" in x[0])]
            if not stage:
                return None
            out[key] = sum(x[1] for x in stage) / steps
            out[key + "_kernels"] = {x[0][:48]: [x[1] / steps, x[2]] for x in stage}          # per launch, dispatches in the pass
            try:                                       # the pass's own bench line: the work it did must be the parent's (same reads, same index)
                cl = json.loads(r.stdout.decode(errors="replace").strip().splitlines()[-1])
                out[key + "_pass_work"] = {k: cl["config"].get(k) for k in ("smems_per_read", "hits_per_read", "searches_per_read")} | {"stage_ms": cl["roofline"].get("kernel_ms")}
            except Exception:
                pass
            pw = out.get(key + "_pass_work")
            if expect and pw and any(abs((pw.get(k) or 0) - v) > 1e-6 * max(abs(v), 1.0) for k, v in expect.items()):
                log("pmc pass %s did other work than this run (%r against %r): not used" % (counter, pw, expect))
                return None
        except Exception as e:
            log("pmc pass %s failed: %r" % (counter, e))
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out


def sam_md5(path):
    """md5 and line count of a SAM file without its @PG line (it holds the command line).  Header lines are read one by one, the records in blocks."""
    import hashlib
    h = hashlib.md5()
    nlines = 0
    with open(path, "rb") as fh:
        while True:
            pos = fh.tell()
            line = fh.readline()
            if not line.startswith(b"@"):
                fh.seek(pos)
                break
            if not line.startswith(b"@PG"):
                h.update(line)
                nlines += 1
        while True:
            blk = fh.read(1 << 24)
            if not blk:
                break
            h.update(blk)
            nlines += blk.count(b"\n")
    return h.hexdigest(), nlines


class RefCache:
    """Results of the compiled reference's legs (minutes each: it expands its index on the host every time), kept in /dev/shm so that the
    N=2,4,8 runs of a scaling sweep on the same box reuse what the N=1 run measured instead of timing the same CPU job again."""
    def __init__(self, l_pac, bits):
        root = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
        self.path = os.path.join(root, "meme_bench_refcache_%d_b%d.json" % (l_pac, bits))

    def _load(self):
        try:
            return json.load(open(self.path))
        except (OSError, ValueError):
            return {}

    def get(self, key):
        return self._load().get(key)

    def put(self, key, val):
        d = self._load()
        d[key] = val
        try:
            json.dump(d, open(self.path, "w"))
        except OSError:
            pass


def wait_for_free_gpus(n, timeout=120.0):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if all(torch.cuda.mem_get_info(d)[0] > 0.9 * torch.cuda.mem_get_info(d)[1] for d in range(1, min(n, torch.cuda.device_count()))):
            return
        time.sleep(1.0)


REF_BEST_THREADS = int(os.environ.get("MEME_BENCH_REF_THREADS", "64"))
MALLOC_TUNABLES = "glibc.malloc.tcache_count=4000:glibc.malloc.trim_threshold=1073741824:glibc.malloc.top_pad=67108864:glibc.malloc.mmap_threshold=33554432"


def e2e_leg(prefix, genome, npairs, threads, devices=1, refcache=None, read_len=READ_LEN, sub=0.01, indel=0.0, seed=5, slices_ok=True):
    """BASELINE.json's end-to-end metric: `mem -7` on paired-end reads through the reference aligner with the HIP
    backend bound in (oracle/_ref/bwa-meme_dropin = reference main + reference objects + bwa-meme_amd/binding) and
    through the unmodified reference (oracle/_ref/bwa-meme_mode3, AVX-512) on the same host cores, same index files,
    same FASTQ; SAM files compared by md5 (minus @PG).  Walls include index loading; `process_s` sums the reference's own
    per-chunk "Processed N reads in X real sec" lines (seeding + chaining + extension + SAM formation, no I/O).
    Defaults: BASELINE configs[2]'s reads (150 bp, 1 % substitutions); configs[4]'s class passes read_len 250, sub 0.05, indel 0.0075."""
    import re
    ref_dir = os.path.join(REPO, "oracle", "_ref")
    # (a probe may name several builds of the bound aligner, e.g. "bwa-meme_dropin,bwa-meme_dropin_prof" -- the second with SAM-phase timers:
    # the first is the one reported, the others land under "extra_runs")
    # (an entry may carry environment settings of its own: "bwa-meme_dropin@MEME_DROPIN_CIGAR=0"; a path below oracle/_ref is allowed:
    # "r04/bwa-meme_dropin_r04", a build of an earlier round with its own libraries beside it, for an A/B on one box)
    dropin_exes = os.environ.get("MEME_BENCH_E2E_DROPIN_EXE", "bwa-meme_dropin").split(",")
    dropin_exe = dropin_exes[0]
    skip_ref = os.environ.get("MEME_BENCH_E2E_SKIP_REF") == "1"                       # probes of the bound aligner alone: sam_identical is then null
    for exe in ["bwa-meme_mode3"] + dropin_exes:
        if not os.path.exists(os.path.join(ref_dir, exe.split("@")[0])):
            raise RuntimeError("%s not built" % exe)
    d = tempfile.mkdtemp(prefix="meme_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        rng = np.random.default_rng(seed)
        fqs = [os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")]
        t_gen = time.time()
        step = 1 << 20
        for p0 in range(0, npairs, step):                        # pairs are sampled and written a million at a time
            m = min(step, npairs - p0)
            r1, r2 = workload.make_pairs_chunk(genome, m, read_len, rng, sub, indel)
            workload.write_fastq_fast(fqs[0], r1, prefix="p", first=p0, append=p0 > 0)
            workload.write_fastq_fast(fqs[1], r2, prefix="p", first=p0, append=p0 > 0)
            del r1, r2
        log("e2e: %d pairs of %d-bp reads written in %.1f s" % (npairs, read_len, time.time() - t_gen))
        out = {}
        ckey = "e2e_reference_%d_%d_t%d_%s" % (npairs, read_len, threads, os.path.basename(os.path.dirname(prefix)))
        # (an N=1 run times the reference itself -- its line must not lean on another run's baseline -- unless a probe that launches the
        # bound aligner several times on one box asks for the entry explicitly: MEME_BENCH_E2E_REUSE_REF=1; the object then says "cached")
        reuse = devices > 1 or os.environ.get("MEME_BENCH_E2E_REUSE_REF") == "1"
        cached = refcache.get(ckey) if (refcache is not None and reuse) else None
        for exe in ["bwa-meme_mode3"] + dropin_exes:
            if exe == "bwa-meme_mode3" and cached:
                out[exe] = dict(cached, cached="timed by the N=1 run on this box")
                continue
            if exe == "bwa-meme_mode3" and skip_ref:
                continue
            sam = os.path.join(d, re.sub(r"[^A-Za-z0-9_.=-]", "_", exe) + ".sam")
            env = dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_VERBOSE="1", MEME_DROPIN_DEVICES=str(devices))
            spec, exe = exe, exe.split("@")[0]
            for kv in spec.split("@")[1:]:
                env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
            # Both binaries run on glibc malloc (the reference's default build would link mimalloc, a submodule that is not in the tree), and
            # the SAM phase is allocator-bound: both get the same allocator settings -- freed memory stays in the arenas, a deep per-thread
            # cache (profiles/r04_e2e.md: 2.4 -> 1.3 s of mem_process_seqs per 4 M reads for the drop-in)
            env.setdefault("GLIBC_TUNABLES", MALLOC_TUNABLES)
            if torch.cuda.device_count() < devices:      # fewer GPUs than asked for (the 1-GPU test box): device slots share the GPUs
                env["MEME_DROPIN_VIRTUAL"] = str(devices)
            t0 = time.time()
            with open(sam, "wb") as fh:
                # both binaries at the thread count that is best for them on this box -- 64 for either: the unmodified reference's
                # mem_process_seqs takes 11.86 / 11.86 / 11.95 / 13.64 s at 32 / 64 / 128 / 256 threads (2 M pairs, 512 Mbp,
                # profiles/r04_ref_thread_sweeps.md), the drop-in's SAM phase stops scaling at 32
                nthr = min(threads, REF_BEST_THREADS) if exe == "bwa-meme_mode3" else min(threads, int(env.get("MEME_T", os.environ.get("MEME_BENCH_E2E_DROPIN_THREADS", "64"))))   # (a probe entry may say "@MEME_T=24")
                r = subprocess.run([os.path.join(ref_dir, exe), "mem", "-7", "-Y", "-K", "100000000", "-t", str(nthr),
                                    prefix] + fqs, stdout=fh, stderr=subprocess.PIPE, env=env, timeout=3000)
            wall = time.time() - t0
            err = r.stderr.decode(errors="replace")
            if os.environ.get("MEME_BENCH_E2E_STDERR"):            # keep the aligner's own profile / the binding's per-chunk report
                os.makedirs(os.environ["MEME_BENCH_E2E_STDERR"], exist_ok=True)
                open(os.path.join(os.environ["MEME_BENCH_E2E_STDERR"], "%s_%dbp.stderr" % (re.sub(r"[^A-Za-z0-9_.=-]", "_", spec), read_len)), "w").write(err)
            if r.returncode != 0:
                raise RuntimeError("%s failed: %s" % (exe, err[-800:]))
            proc = sum(float(m.group(1)) for m in re.finditer(r"Processed \d+ reads in [0-9.]+ CPU sec, ([0-9.]+) real sec", err))
            proc_cpu = sum(float(m.group(1)) for m in re.finditer(r"Processed \d+ reads in ([0-9.]+) CPU sec", err))
            md5, nlines = sam_md5(sam)
            # (probes of run-to-run stability: MEME_BENCH_E2E_KEEP_DIFF=1 keeps the first SAM file and writes the records in which a later
            # run differs from it next to the aligners' stderr)
            if os.environ.get("MEME_BENCH_E2E_KEEP_DIFF") and os.environ.get("MEME_BENCH_E2E_STDERR"):
                first_sam = os.path.join(d, "first.sam")
                if not os.path.exists(first_sam):
                    os.rename(sam, first_sam)
                    open(sam, "wb").close()
                else:
                    nd = 0
                    with open(first_sam, "rb") as fa, open(sam, "rb") as fb, open(os.path.join(os.environ["MEME_BENCH_E2E_STDERR"], "diff_%s.txt" % re.sub(r"[^A-Za-z0-9_.=-]", "_", spec)), "wb") as fo:
                        for la, lb in zip(fa, fb):
                            if la != lb and not la.startswith(b"@PG"):
                                nd += 1
                                if nd <= 200:
                                    fo.write(b"< " + la + b"> " + lb)
                    log("e2e: %s differs from the first run in %d records" % (spec, nd))
            os.remove(sam)
            info = {"threads": nthr, "wall_s": wall, "process_s": proc, "process_cpu_s": proc_cpu, "reads_per_s_wall": 2 * npairs / wall,
                    "reads_per_s_process": 2 * npairs / proc if proc > 0 else None, "sam_md5": md5, "sam_lines": nlines}
            m = re.search(r"Runtime-build-index took ([0-9.]+) sec", err)
            if m:
                info["host_index_expansion_s"] = float(m.group(1))
            m = re.search(r"index staged in HBM in ([0-9.]+) s", err)
            if m:
                info["hbm_index_staging_s"] = float(m.group(1))
            for m in re.finditer(r"totals: chunk-level device stages \(gather \+ seeding \+ chaining \+ extension\) ([0-9.]+) s for (\d+) reads, of which the seeding calls "
                                 r"([0-9.]+) s; bsw (\d+) calls, (\d+) pairs \(copy-in thread-seconds ([0-9.]+), backend calls ([0-9.]+) s of which kernels ([0-9.]+) s\)", err):
                # device_stages_s: everything the binding does for a chunk ahead of the reference's own body -- gathering the reads, the seeding
                # call, the chaining + extension call (extension_stage_s below); seeding_call_s: the seeding call alone (H2D + kernels)
                info["backend"] = {"device_stages_s": float(m.group(1)), "seeding_call_s": float(m.group(3)), "bsw_calls": int(m.group(4)),
                                   "bsw_pairs": int(m.group(5)), "bsw_backend_s": float(m.group(7)), "bsw_kernel_s": float(m.group(8))}
            for m in re.finditer(r"extension: this chunk .*?totals ([0-9.]+) s, (\d+) backend calls", err):
                info.setdefault("backend", {})["extension_stage_s"] = float(m.group(1))      # host stage (MEME_DROPIN_EXT=host): jobs built + calls + fold + purge
            for m in re.finditer(r"CIGAR stage on the device: (\d+) alignments posed so far \(kernels ([0-9.]+) s, whole pre-pass ([0-9.]+) s\); "
                                 r"bwa_gen_cigar2 calls answered from the table (\d+), computed by the reference's function (\d+)", err):
                info.setdefault("backend", {}).update({"cigar_jobs": int(m.group(1)), "cigar_kernels_s": float(m.group(2)), "cigar_prepass_s": float(m.group(3)),
                                                        "cigar_calls_from_table": int(m.group(4)), "cigar_calls_by_reference": int(m.group(5))})
            for m in re.finditer(r"mate rescue on the device: (\d+) Smith-Waterman jobs posed so far \(kernels ([0-9.]+) s, whole pre-pass ([0-9.]+) s\)", err):
                info.setdefault("backend", {}).update({"mate_rescue_jobs": int(m.group(1)), "mate_rescue_kernels_s": float(m.group(2)), "mate_rescue_prepass_s": float(m.group(3))})
            for m in re.finditer(r"chaining \+ extension on the device: ([0-9.]+) s in the backend calls so far \(HIP events: chaining ([0-9.]+) s, extension stage "
                                 r"([0-9.]+) s of which banded SW ([0-9.]+) s\); (\d+) alignment records, (\d+) extension jobs \((\d+) of them again with the doubled "
                                 r"band\), (\d+) reads chained by the wavefront-per-read tier, (\d+) reads chained on the host", err):
                info.setdefault("backend", {}).update({
                    "extension_stage_s": float(m.group(1)),     # wall of the chaining + extension calls (kernels, their host round trips, D2H of the records)
                    "chain_kernels_s": float(m.group(2)), "ext_kernels_s": float(m.group(3)), "bsw_kernel_s": float(m.group(4)),
                    "alignment_records": int(m.group(5)), "bsw_pairs": int(m.group(6)), "bsw_pairs_doubled_band": int(m.group(7)),
                    "reads_chained_on_host": int(m.group(9))})
            for m in re.finditer(r"SAM text on the device: (\d+) records formatted there so far \(kernels ([0-9.]+) s, whole stage ([0-9.]+) s\), (\d+) by the reference's mem_aln2sam", err):
                info.setdefault("backend", {}).update({"sam_records_on_device": int(m.group(1)), "sam_kernels_s": float(m.group(2)), "sam_stage_s": float(m.group(3)),
                                                        "sam_records_by_reference": int(m.group(4))})
            prof = re.findall(r"\[meme-dropin-prof\]   (.+?)\s+([0-9.]+) s\s+(\d+) calls", err)
            if prof:
                info["sam_phase_thread_seconds"] = {k.strip(): {"s": float(v), "calls": int(c)} for k, v, c in prof}
            out[spec] = info
            if exe == "bwa-meme_mode3" and refcache is not None:
                refcache.put(ckey, info)
            log("e2e: %s wall %.1f s, process %.1f s (CPU %.1f s), %d SAM lines" % (spec, wall, proc, proc_cpu, nlines))
        # ---- the device-stage floor per slice (round 6): with N GPUs the binding hands every GPU 1/N of a -K chunk, so what an N-GPU run's
        # device stages take is decided by how the stages of a 667 k / N-read slice scale.  The bound aligner once more per slice size (-K = slice x read length,
        # one GPU, no reference run): seconds of device stages per slice -> the predicted device-stage wall of this workload at 2, 4, 8 GPUs.
        slices = None
        s_pairs = npairs
        if os.environ.get("MEME_BENCH_E2E_SLICES", "1") != "0" and devices == 1 and read_len == READ_LEN and slices_ok:
            slices = {}
            chunk_reads = 100000000 // read_len // 2 * 2 + 2
            # (on the first four chunks' worth of pairs: per-slice figures do not depend on how many chunks follow, and three more runs over all 10 M pairs were 50 s of this leg)
            s_pairs = min(npairs, int(os.environ.get("MEME_BENCH_E2E_SLICE_PAIRS", "2666668")))
            s_fqs = fqs
            if s_pairs < npairs:
                s_fqs = [os.path.join(d, "s1.fq"), os.path.join(d, "s2.fq")]
                for src, dst in zip(fqs, s_fqs):
                    with open(dst, "wb") as fh:
                        subprocess.run(["head", "-n", str(4 * s_pairs), src], stdout=fh, check=True)
            for div in (8, 4, 2):
                k_bases = 100000000 // div
                env = dict(os.environ, MEME_INDEX_PREFIX=prefix, MEME_DROPIN_VERBOSE="1", MEME_DROPIN_DEVICES="1")
                env.setdefault("GLIBC_TUNABLES", MALLOC_TUNABLES)
                t0 = time.time()
                r = subprocess.run([os.path.join(ref_dir, dropin_exe.split("@")[0]), "mem", "-7", "-Y", "-K", str(k_bases), "-t", str(min(threads, 64)), prefix] + s_fqs,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env, timeout=1200)
                err = r.stderr.decode(errors="replace")
                mm = re.findall(r"totals: chunk-level device stages \(gather \+ seeding \+ chaining \+ extension\) ([0-9.]+) s for (\d+) reads", err)
                pr = sum(float(m.group(1)) for m in re.finditer(r"Processed \d+ reads in [0-9.]+ CPU sec, ([0-9.]+) real sec", err))
                if r.returncode == 0 and mm:
                    dev_s, nr = float(mm[-1][0]), int(mm[-1][1])
                    slices[str(div)] = {"slice_reads": chunk_reads // div, "pairs_of_the_run": s_pairs, "device_stages_s_total": dev_s, "device_stages_ms_per_slice": 1e3 * dev_s / max(nr, 1) * (chunk_reads // div),
                                        "process_s": pr, "wall_s": time.time() - t0}
            log("e2e: device stages per slice of 1/8, 1/4, 1/2 chunk: %s" % {k: round(v["device_stages_ms_per_slice"], 1) for k, v in slices.items()})
        ref, drop = out.get("bwa-meme_mode3"), out[dropin_exe]
        # what bounds the bound aligner (for reading an N-GPU curve: the device stages of a chunk run on its GPUs side by side -- one slice each -- and,
        # from the second chunk on, beside the previous chunk's host phases; mem_process_seqs is the host's own time)
        dev_s = drop.get("backend", {}).get("device_stages_s")
        if dev_s is not None and drop["process_s"] > 0:
            drop["bound"] = {"device_stages_wall_s": dev_s, "device_stages_wall_s_per_gpu_if_split_evenly": dev_s, "gpus": devices, "host_process_s": drop["process_s"],
                             "host_process_cpu_s": drop["process_cpu_s"], "host_cpu_quota": host_cpu_quota(),
                             "device_stage_floor_per_slice": (dict(slices, **{"1": {"slice_reads": 100000000 // read_len // 2 * 2 + 2, "device_stages_s_total": dev_s,
                                                                                  "device_stages_ms_per_slice": 1e3 * dev_s / (2.0 * npairs) * (100000000 // read_len // 2 * 2 + 2)}})
                                                              if slices else None),
                             "predicted_device_stages_wall_s_by_gpus": ({g: (slices[g]["device_stages_s_total"] * (npairs / float(s_pairs)) / int(g) if g in slices else None) for g in ("2", "4", "8")} | {"1": dev_s}
                                                                        if slices else None),
                             "prediction": "N GPUs: every GPU takes 1/N of a chunk; the device-stage wall of this workload is then (seconds per slice of that size) x (number of chunks) = "
                                           "the one-GPU total at -K/N divided by N",
                             "reading": ("host-bound: mem_process_seqs (%.2f s) exceeds the device stages (%.2f s wall, every GPU working on its slice at once); more GPUs "
                                         "shorten only the device stages" % (drop["process_s"], dev_s)) if drop["process_s"] >= dev_s else
                                        ("device-bound: the device stages (%.2f s) exceed mem_process_seqs (%.2f s); more GPUs shorten the run" % (dev_s, drop["process_s"]))}
        return {"metric": "e2e_reads_per_sec", "value": drop["reads_per_s_wall"], "unit": "reads/s",
                "workload": "mem -7 (reference: -t %s, its best of a 32-256 sweep; with the backend bound: -t %d), %d pairs of %d-bp reads (%g %% substitutions, %g %% indels, "
                            "300-500 bp inserts%s) vs the benchmark genome (%d bp), wall time incl. index loading"
                            % (ref["threads"] if ref else "-", drop["threads"], npairs, read_len, 100 * sub, 100 * indel, " + %d" % (read_len - 150) if read_len != 150 else "", genome.shape[0]),
                "allocator": "both binaries on glibc malloc with GLIBC_TUNABLES=%s; the bound aligner also calls mallopt (trim threshold 1 GB, top pad 64 MB, "
                             "mmap threshold 32 MB) and buffers stdout (16 MB)" % MALLOC_TUNABLES,
                "threads": threads, "pairs": npairs, "read_len": read_len, "gpus_driven_by_the_one_aligner_process": devices,
                "sam_identical": bool(ref["sam_md5"] == drop["sam_md5"]) if ref else None,
                "extra_runs_sam_identical_to_the_first": all(out[e]["sam_md5"] == drop["sam_md5"] for e in dropin_exes[1:]) if len(dropin_exes) > 1 else None,
                "dropin": drop, "reference": ref, "extra_runs": {e: out[e] for e in dropin_exes[1:]} or None, "speedup_wall": ref["wall_s"] / drop["wall_s"] if ref else None,
                "speedup_process": (ref["process_s"] / drop["process_s"]) if ref and drop["process_s"] > 0 else None}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: re-executes itself under torch.distributed.run, one rank per GPU over RCCL.
    On a box with fewer GPUs than ranks (the 1-GPU test box) the ranks share the devices and talk over gloo -- RCCL refuses two
    ranks on one device; the line then says so (`collective.backend`, `collective.ranks_per_device`)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    if torch.cuda.device_count() < a.gpus:
        env.setdefault("MEME_BENCH_BACKEND", "gloo")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(a.gpus), "--steps", str(a.steps), "--warmup", str(a.warmup)]
    log("spawning %d ranks: %s" % (a.gpus, " ".join(cmd[1:])))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MEME_BENCH_DEVICE"):        # testing the multi-rank path on a box with fewer GPUs than ranks (with MEME_BENCH_BACKEND=gloo)
        local = int(os.environ["MEME_BENCH_DEVICE"])
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP backend has no CPU fallback")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))                   # `python bench.py --gpus N` spawns its own ranks (one per GPU)
    if a.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU" % (a.gpus, world))
    n_dev = torch.cuda.device_count()
    ranks_per_device = 1
    if local >= n_dev:                             # more ranks than GPUs (tests on a 1-GPU box): ranks share devices, gloo instead of RCCL
        ranks_per_device = (world + n_dev - 1) // n_dev
        local = local % n_dev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # rank 0 builds the index on its host for several minutes while the others wait at the first broadcast
        backend = os.environ.get("MEME_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=90))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(minutes=90))

    mbp = float(os.environ.get("MEME_BENCH_MBP", "3100"))
    nreads = int(os.environ.get("MEME_BENCH_READS", "10000000"))
    bits = int(os.environ.get("MEME_BENCH_BITS", "0"))
    # host RAM / HBM guards, decided by rank 0 for everybody: the builder needs ~45 B of host RAM per suffix,
    # the GPU ~28 B per suffix plus the batch; shrink the genome rather than die
    if rank == 0:
        try:
            import psutil
            avail = psutil.virtual_memory().available
            while mbp > 64 and 2 * mbp * 1e6 * 45 > 0.8 * avail:
                mbp /= 2
                log("host RAM too small for the configured genome: falling back to %.0f Mbp" % mbp)
        except ImportError:
            pass
        gpu_free = torch.cuda.mem_get_info(local)[0] // ranks_per_device
        while mbp > 64 and 2 * mbp * 1e6 * 33 + nreads * 3800 > 0.9 * gpu_free:     # (3 KB of SMEM slots + packed read + outputs per read)
            if os.environ.get("MEME_BENCH_NO_FALLBACK") == "1":
                # (the counter passes re-execute this file while their parent is still alive: found in round 6 -- what the parent had not freed made every pass of
                # rounds 5-6 fall back to HALF the genome without a word, and `roofline.traffic` was that configuration's; a pass now stops instead)
                log("HBM too small for the configured genome (%.0f GB free) and MEME_BENCH_NO_FALLBACK=1: stopping" % (gpu_free / 1e9))
                sys.exit(3)
            mbp /= 2
            log("HBM too small for the configured genome: falling back to %.0f Mbp" % mbp)
    l_pac_t = torch.tensor([int(mbp * 1e6) & ~1], dtype=torch.int64, device=dev)
    if world > 1:
        dist.broadcast(l_pac_t, 0)
    l_pac = int(l_pac_t[0])
    n = 2 * l_pac

    # ---- index: built on rank 0's host, staged on every GPU ----------------------------------------
    meta = torch.zeros(4, dtype=torch.int64, device=dev)
    fwd = text = sa = l1 = l2 = pre = None
    ctx = hipapi.Context(local)
    if rank == 0:
        t0 = time.time()
        fwd = synth.make_genome(l_pac, seed=11)
        cache = None
        # (only the host build is worth caching: the device build is faster than reading 62 GB back from /dev/shm)
        if os.environ.get("MEME_BENCH_CACHE", "1") != "0" and os.path.isdir("/dev/shm") and os.environ.get("MEME_BENCH_SA", "device") != "device":
            cache = "/dev/shm/meme_bench_idx_%d_b%d" % (l_pac, bits)
        if cache and os.path.exists(cache + ".ok"):
            text = np.fromfile(cache + ".text", dtype=np.uint8)
            sa = np.fromfile(cache + ".sa", dtype=np.uint64)
            l1 = np.fromfile(cache + ".l1", dtype=hostapi.RMI_DTYPE)
            l2 = np.fromfile(cache + ".l2", dtype=hostapi.RMI_DTYPE)
            log("genome %.0f Mbp: index loaded from the /dev/shm cache in %.1f s" % (l_pac / 1e6, time.time() - t0))
        else:
            if os.environ.get("MEME_BENCH_SA", "device") == "device":
                # the whole index on the GPU: suffix array (meme_sa_build_device: radix sort + prefix doubling), the 5-byte
                # image, the probe-ready entries and the P-RMI (meme_prmi_train_device); the host keeps copies for the
                # reference-format index files of the cpu_baseline / e2e legs
                text = hipapi.fwd_rc_text(fwd)
                t_gen = time.time() - t0
                d_text0, d_s = hipapi.build_sa_device(ctx, text)
                t_sa = time.time() - t0 - t_gen
                d_pos50 = hipapi.pos5_from_sa_torch(ctx, d_s, n)
                sa = d_s.cpu().numpy().view(np.uint64)
                del d_s
                torch.cuda.empty_cache()
                t1 = time.time()
                d_pac0, d_ent0 = hipapi.stage_entries_torch(ctx, n, d_text0, d_pos50)
                use_bits = bits if bits > 0 else (28 if 8.0 * n + 8 > 8.0e9 else 26 if 8.0 * n + 8 > 1.0e9 else 24)   # build_rmis_dna.sh:68-77
                d_l2_0, n_l2_0, d_l1_0, n_l1_0 = hipapi.train_prmi_device(ctx, d_ent0, n, use_bits)
                t_train = time.time() - t1
                l2 = d_l2_0.cpu().numpy().view(hostapi.RMI_DTYPE)
                l1 = d_l1_0.cpu().numpy().view(hostapi.RMI_DTYPE)[:n_l1_0]
                pre = (d_text0, d_pos50, d_l2_0, d_l1_0, d_pac0, d_ent0)
                log("genome %.0f Mbp synthesised in %.1f s; on the device: suffix array in %.1f s (incl. the upload of the text), "
                    "entries + P-RMI (2^%d leaves, %d partial) in %.1f s" % (l_pac / 1e6, t_gen, t_sa, use_bits, n_l1_0, t_train))
            else:
                text, sa = hostapi.build_sa(fwd)
                t_sa = time.time() - t0
                l1, l2 = hostapi.train_prmi(text, sa, bits=bits)
                log("genome %.0f Mbp: suffix array in %.1f s (host), P-RMI (2^%d leaves, %d partial) trained on the host in %.1f s"
                    % (l_pac / 1e6, t_sa, int(np.log2(l2.shape[0])), l1.shape[0], time.time() - t0 - t_sa))
            if cache:
                try:
                    import glob
                    for old in glob.glob("/dev/shm/meme_bench_idx_*"):
                        if not old.startswith(cache + "."):
                            os.remove(old)
                    if shutil.disk_usage("/dev/shm").free > 1.2 * (text.nbytes + sa.nbytes + l1.nbytes + l2.nbytes):
                        text.tofile(cache + ".text"); sa.tofile(cache + ".sa"); l1.tofile(cache + ".l1"); l2.tofile(cache + ".l2")
                        open(cache + ".ok", "w").write("ok")
                except OSError as e:
                    log("index cache not written: %r" % (e,))
                    for ext in (".text", ".sa", ".l1", ".l2", ".ok"):
                        try:
                            os.remove(cache + ext)
                        except OSError:
                            pass
        meta[0], meta[1], meta[2] = n, l2.shape[0], l1.shape[0]
    if world > 1:
        dist.broadcast(meta, 0)
    n_l2, n_l1 = int(meta[1]), int(meta[2])
    t0 = time.time()
    if os.environ.get("MEME_BENCH_LANES"):
        ctx.set_tuning("group_lanes", int(os.environ["MEME_BENCH_LANES"]))
    if os.environ.get("MEME_BENCH_BPC"):
        ctx.set_tuning("seed_blocks_per_cu", int(os.environ["MEME_BENCH_BPC"]))
    if os.environ.get("MEME_BENCH_SMEM_CAP"):
        ctx.set_tuning("smem_cap", int(os.environ["MEME_BENCH_SMEM_CAP"]))
    L = hipapi.lib()
    bcast_bytes = 0
    if pre is not None:
        d_text, d_pos5, d_l2, d_l1 = pre[:4]
    else:
        d_text = torch.empty(n, dtype=torch.uint8, device=dev)
        d_pos5 = torch.zeros(L.meme_index_pos5_bytes(n), dtype=torch.uint8, device=dev)
        d_l2 = torch.empty(n_l2 * 24, dtype=torch.uint8, device=dev)
        d_l1 = torch.empty(max(n_l1, 1) * 24, dtype=torch.uint8, device=dev)
        if rank == 0:
            d_text.copy_(torch.from_numpy(text))
            # the host builder holds the suffix array as u64; the GPU index (and the broadcast) use the reference's 5-byte image
            d_sa = torch.from_numpy(sa.view(np.int64)).to(dev)
            d_pos5 = hipapi.pos5_from_sa_torch(ctx, d_sa, n)
            del d_sa
            torch.cuda.empty_cache()
            d_l2.copy_(torch.from_numpy(l2.view(np.uint8).reshape(-1)))
            if n_l1:
                d_l1[:n_l1 * 24].copy_(torch.from_numpy(l1.view(np.uint8).reshape(-1)))
    t_bcast = 0.0
    if world > 1:
        # one-off RCCL broadcast of the raw index images over xGMI (5 B per suffix + 1 B per base + the model tables);
        # no collective in steady state
        torch.cuda.synchronize()
        tb0 = time.time()
        for t in (d_text, d_pos5, d_l2, d_l1):
            dist.broadcast(t, 0)
            bcast_bytes += t.numel() * t.element_size()
        torch.cuda.synchronize()
        t_bcast = time.time() - tb0
    torch.cuda.synchronize()
    t_stage0 = time.time()
    if pre is not None:
        keep = (pre[4], pre[5]) + hipapi.attach_index_torch(ctx, n, pre[4], pre[5], d_l2, n_l2, d_l1, n_l1)
    else:
        keep = hipapi.stage_index_torch(ctx, n, d_text, d_pos5, d_l2, n_l2, d_l1, n_l1)
    del d_text, d_l2, d_l1, d_pos5
    pre = None
    torch.cuda.empty_cache()
    t_stage = time.time() - t_stage0                 # this rank's staging kernels (entries, model records, plcp table) behind the broadcast
    log("index staged in HBM in %.1f s (%.2f GB entries; broadcast %.1f s, staging kernels %.1f s)" % (time.time() - t0, 16 * n / 1e9, t_bcast, t_stage))

    # ---- reads: every rank samples its own batch ------------------------------------------------------
    genome_t = torch.empty(l_pac, dtype=torch.uint8, device=dev)
    if rank == 0:
        genome_t.copy_(torch.from_numpy(fwd))
    if world > 1:
        dist.broadcast(genome_t, 0)
    genome = fwd if rank == 0 else genome_t.cpu().numpy()
    del genome_t
    t0 = time.time()
    rf = os.environ.get("MEME_BENCH_READS_FILE")          # (the counter passes re-execute this file: they take the parent's batch instead of sampling it again)
    if rf and os.path.exists(rf) and world == 1:
        reads = np.load(rf)
        assert reads.shape == (nreads, READ_LEN)
    else:
        reads = workload.make_reads_fast(genome, nreads, READ_LEN, seed=1000 + rank)
    d_reads = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_off = torch.arange(0, (nreads + 1) * READ_LEN, READ_LEN, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    log("%d reads sampled and uploaded in %.1f s" % (nreads, time.time() - t0))
    # the batch as sampled, for the counter passes at the end of the run (they re-execute this file; sampling again was 18 s per pass)
    reads_file, reads_sum0 = None, int(reads.reshape(-1).view(np.uint64).sum(dtype=np.uint64)) if reads.size % 8 == 0 else None
    if os.environ.get("MEME_BENCH_KEEP_READS"):                  # (a probe: the batch as a file of its own)
        np.save(os.environ["MEME_BENCH_KEEP_READS"], reads)
    if rank == 0 and world == 1 and not rf and os.environ.get("MEME_BENCH_PMC", "1") != "0" and os.environ.get("MEME_BENCH_PMC_READS_CACHE", "0") != "0" and os.path.isdir("/dev/shm") and shutil.which("rocprofv3"):
        try:
            reads_file = os.path.join("/dev/shm", "meme_bench_reads_%d.npy" % os.getpid())
            np.save(reads_file, reads)
            import atexit
            atexit.register(lambda f=reads_file: os.path.exists(f) and os.remove(f))
        except Exception:
            reads_file = None
    opt = hipapi.default_seed_opt(rounds=3)

    def step():
        return ctx.seed_batch_device(d_reads.data_ptr(), d_off.data_ptr(), nreads, nreads * READ_LEN, opt)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    for _ in range(a.warmup):
        res = step()
    kernel_ms = []
    windows = 0
    rc_exit = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = step()
        tm = ctx.timings()
        kernel_ms.append((tm.seed_kernel_ms, tm.seed_gather_ms + tm.seed_pack_ms, tm.seed_reseed_ms))
        windows = tm.seed_windows
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt[0])
    # what every rank processed (whole-job value = all ranks' reads over the slowest rank's time), and what the collective layer saw
    per_rank = [nreads * a.steps]
    collective = {"backend": None, "ranks_seen": 1, "ranks_per_device": 1, "index_broadcast_bytes": 0}
    if world > 1:
        cnt = torch.tensor([nreads * a.steps], dtype=torch.int64, device=dev)
        got = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(got, cnt)
        per_rank = [int(g[0]) for g in got]
        # what a reader of the N > 1 line needs to tell start-up from steady state: the broadcast, every rank's staging and kernel times
        mine = torch.tensor([t_bcast, t_stage, float(np.mean([k[0] for k in kernel_ms])), float(np.mean([k[1] for k in kernel_ms])), time.time() - T_START],
                            dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        collective = {"backend": "rccl" if dist.get_backend() == "nccl" else dist.get_backend(), "ranks_seen": dist.get_world_size(),
                      "ranks_per_device": ranks_per_device, "index_broadcast_bytes": int(bcast_bytes),
                      "index_broadcast_s": max(float(x[0]) for x in allr),
                      "index_broadcast_GBps": (bcast_bytes / 1e9 / max(float(x[0]) for x in allr)) if max(float(x[0]) for x in allr) > 0 else None,
                      "per_rank": [{"staging_s": float(x[1]), "search_stage_ms": float(x[2]), "pack_gather_ms": float(x[3]), "startup_s": float(x[4])} for x in allr],
                      "startup_budget": "start-up = genome + index build on rank 0, broadcast, staging on every rank: %.0f s on the slowest rank before the timed "
                                        "region of %d steps" % (max(float(x[4]) for x in allr), a.steps)}
        dist.barrier()
        dist.destroy_process_group()                 # no collective in steady state, none after the timed region either
        if rank != 0:                                # rank 0 goes on alone with the reported extras (baseline, legs)
            del keep, d_reads, d_off
            ctx.close()
            sys.exit(0)

    if rank == 0:
        k_ms = float(np.mean([k[0] for k in kernel_ms]))
        g_ms = float(np.mean([k[1] for k in kernel_ms]))
        rs_ms = float(np.mean([k[2] for k in kernel_ms]))
        sample = reads[:20000]
        bpr, per_read = algorithmic_bytes_per_read(text, sa, l1, l2, sample)
        # parity at the benchmark's own size (index of n suffixes), by CONTENT: the GPU seeds the sample on its own and its
        # full seed dump (every SMEM, every hit position, the harness's dump format) must equal the pinned oracle's
        # (oracle/meme_oracle.c orc_seed_batch; the checker only -- nothing timed goes through it)
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import oracle_py as O
        # (round 4: 1 M reads instead of 20 000, in slices the oracle's fixed-capacity output arrays can hold; every SMEM and every hit
        # position compared after the dump format's ordering -- SMEMs by (start asc, end desc), hits in suffix-array order)
        ns = min(nreads, int(os.environ.get("MEME_BENCH_PARITY_READS", "600000")))
        t_par = time.time()
        sample_parity, o_idx = True, O.Index(text, sa)
        SL = 50000
        for p0 in range(0, ns, SL):
            part = reads[p0:p0 + SL]
            if not seeds_equal_oracle(ctx, O, o_idx, part, opt):
                sample_parity = False
                log("PARITY MISMATCH in reads %d..%d of the benchmark's batch: the GPU's seeds differ from the oracle's" % (p0, p0 + part.shape[0]))
                break
        log("parity: %d of the benchmark's reads against orc_seed_batch in %.1f s: %s" % (ns, time.time() - t_par, "identical" if sample_parity else "DIFFERENT"))
        achieved = bpr * nreads / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "seeding_reads_per_sec", "value": (sum(per_rank) / dt) if sample_parity else None,
            "unit": "reads/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "learned-index seeding (rounds 1-3 + hit gather), %d bp SE reads vs %.0f Mbp synthetic "
                                   "genome (fwd+rc suffix array of %d entries), index and reads resident in HBM"
                                   % (READ_LEN, l_pac / 1e6, n),
                       "genome_bp": l_pac, "sa_entries": n, "reads_per_gpu_per_step": nreads, "read_len": READ_LEN,
                       "rmi_leaves_log2": int(np.log2(n_l2)), "sharding": "reads/%d ranks, index replicated by RCCL broadcast" % world,
                       "reads_per_rank": per_rank, "collective": collective,
                       "smems_per_read": res.total_smems / nreads, "hits_per_read": res.total_hits / nreads,
                       "searches_per_read": res.searches / nreads,
                       "windows_per_search": windows / max(res.searches, 1),
                       "sample_parity_with_oracle": bool(sample_parity),
                       "sample_parity_check": "every SMEM and hit position of %d of the benchmark's reads vs orc_seed_batch" % ns},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": None, "kernel": "the SA-search stage: k_seed (rounds 1 and 3, re-seeding of repeated SMEMs) + k_reseed / _emit / _search / _resume "
                                                    "(re-seeding of unique SMEMs on the plcp table); HIP events around the whole stage",
                         "kernel_ms": k_ms, "of_which_reseed_kernels_ms": rs_ms, "gather_ms": g_ms,
                         "algorithmic_bytes_per_read": bpr, "work_per_read": per_read},
        }
        # HBM traffic per launch from the committed PMC passes of this very configuration (rocprofv3 counters cannot be
        # collected inside a timed run): FETCH_SIZE doubled as the gfx950 guide prescribes (128-byte line fills are
        # tallied at 64 B; cross-checked against TCC_MISS x 128 B) + WRITE_SIZE
        pmc = None
        try:
            pmc = json.load(open(os.path.join(REPO, "profiles", "pmc_named.json")))
            c = pmc["config"]
            if not (c["genome_bp"] == l_pac and c["reads_per_step"] == nreads and c["read_len"] == READ_LEN and
                    c["rmi_leaves_log2"] == int(np.log2(n_l2))):
                pmc = None
        except (OSError, ValueError, KeyError):
            pmc = None
        if pmc:
            try:
                out["roofline"]["traffic"] = (2 * pmc["fetch_size_kb_per_launch"] + pmc["write_size_kb_per_launch"]) * 1024.0
                out["roofline"]["traffic_source"] = "committed PMC pass, not this run: " + pmc["source"]
                out["roofline"]["l2_miss_lines_per_s"] = pmc["tcc_miss_lines_per_launch"] / (k_ms * 1e-3)
                out["roofline"]["random_line_roofline_lines_per_s"] = 50e9     # scripts/microbench/gather_roofline.hip
            except Exception as e:  # never lose the headline line over an annotation
                log("pmc annotation skipped: %r" % (e,))
        budget = float(os.environ.get("MEME_BENCH_BUDGET_S", "1500"))
        rd_prefix = rd_genome = None
        single = True        # (rank 0 is alone from here on at every N: the other ranks have left after the timed region)
        cores = min(256, os.cpu_count() or 1)
        refcache = RefCache(l_pac, bits)
        # ---- the reference's own file formats on disk, for the two legs that run compiled reference binaries ----------
        ref_prefix = None
        want_ref = single and (os.environ.get("MEME_BENCH_CPU", "reference") == "reference" or
                               os.environ.get("MEME_BENCH_E2E", "1") != "0")
        have_avx512 = "avx512bw" in open("/proc/cpuinfo").read()
        if want_ref and have_avx512 and os.path.exists(os.path.join(REPO, "oracle", "_ref", "learned_seeding_mode3")):
            try:
                ref_prefix = reference_index_on_disk(fwd, text, sa, l1, l2, l_pac, bits)
            except Exception as e:
                log("reference-format index not available: %r" % (e,))
        # ---- cpu_baseline: the compiled reference timed in this run (fallback: the restated port) ------------------------
        cpu = None
        cpu_mode = os.environ.get("MEME_BENCH_CPU", "reference")
        if single and cpu_mode != "0":
            nsamp = min(nreads, int(os.environ.get("MEME_BENCH_CPU_READS", "2000000")))
            cpu = refcache.get("cpu_baseline_%d" % nsamp) if world > 1 else None      # N>1 lines reuse the N=1 run's figure of this box
            if cpu is not None:
                cpu["sample"] += " (timed by the N=1 run on this box, cached in /dev/shm)"
            elif cpu_mode == "reference" and ref_prefix and time.time() - T_START < budget - 400:
                try:
                    # the reference's best thread count on this box: a sweep of the same harness (32 / 64 / 128 / 256 threads: 459 / 485 /
                    # 459 / 447 k reads/s at 512 Mbp, profiles/r04_ref_thread_sweeps.md) -- it stops scaling at 32 threads
                    cpu = cpu_baseline_reference(ref_prefix, reads[:nsamp], min(os.cpu_count() or 1, REF_BEST_THREADS))
                    refcache.put("cpu_baseline_%d" % nsamp, cpu)
                except Exception as e:  # the baseline is a reported extra, never the measured value
                    log("cpu_baseline (reference) failed: %r -- falling back to the port" % (e,))
            port = cpu_baseline_port(text, sa, l1, l2, reads[:min(nsamp, 400000)], os.cpu_count() or 1)
            if cpu is None:
                cpu = port
            else:
                cpu["port"] = {k: port[k] for k in ("value", "unit", "cores", "sample")}
        if cpu is not None:
            # what the host gives this process: the hardware threads it may run on and, where a cgroup limits it, the CPUs' worth of time it
            # gets (the boxes this was developed on: 256 threads visible, 16 by quota) -- every host-side figure of this line runs against that
            cpu["host_threads_visible"] = os.cpu_count() or 1
            cpu["host_cpu_quota"] = host_cpu_quota()
        out["cpu_baseline"] = cpu
        if single and os.environ.get("MEME_BENCH_BSW", "1") != "0":
            try:
                out["bsw"] = bsw_leg(ctx, dev, world)
                log("bsw leg done")
            except Exception as e:  # a secondary measurement: never lose the headline line over it
                log("bsw leg failed: %r" % (e,))
                out["bsw"] = None
        if single and os.environ.get("MEME_BENCH_KSWV", "1") != "0":
            try:
                out["kswv"] = kswv_leg(ctx)
                log("kswv leg done")
            except Exception as e:
                log("kswv leg failed: %r" % (e,))
                out["kswv"] = None
        if single and os.environ.get("MEME_BENCH_CHAIN", "1") != "0":
            try:
                out["chain"] = chain_leg(ctx, reads, l_pac)
                log("chain leg done")
            except Exception as e:
                log("chain leg failed: %r" % (e,))
                out["chain"] = None
        if single and os.environ.get("MEME_BENCH_EXT", "1") != "0":
            try:
                out["ext"] = ext_leg(ctx, reads, fwd, l_pac)
                log("ext leg done")
            except Exception as e:
                log("ext leg failed: %r" % (e,))
                out["ext"] = None
        # ---- BASELINE configs[4]'s read class (250 bp, 5 % substitutions, 0.75 % indels) through the same stages -------------------------------
        if single and os.environ.get("MEME_BENCH_C4", "1") != "0":
            if time.time() - T_START > budget - 700:
                out["config4_class"] = {"skipped": "wall budget (%d s) nearly used up after %.0f s" % (budget, time.time() - T_START)}
            else:
                try:
                    out["config4_class"] = config4_class_leg(ctx, dev, fwd, text, sa, l1, l2, l_pac)
                    log("config4_class leg done")
                except Exception as e:
                    log("config4_class leg failed: %r" % (e,))
                    out["config4_class"] = {"failed": repr(e)[:300]}
        # ---- every stage on a repeat-dense genome (round 6: the headline's genome is 98 % unique; this one carries the overflow tiers, max_occ and the heavy chaining tiers) ----
        if single and os.environ.get("MEME_BENCH_RD", "1") != "0":
            if time.time() - T_START > budget - 650:
                out["repeat_dense"] = {"skipped": "wall budget (%d s) nearly used up after %.0f s" % (budget, time.time() - T_START)}
            else:
                try:
                    out["repeat_dense"] = repeat_dense_leg(local, want_ref_index=bool(ref_prefix) and os.environ.get("MEME_BENCH_E2E", "1") != "0")
                    rd_prefix, rd_genome = out["repeat_dense"].pop("_ref_prefix", None), out["repeat_dense"].pop("_genome", None)
                except Exception as e:
                    log("repeat_dense leg failed: %r" % (e,))
                    out["repeat_dense"] = {"failed": repr(e)[:300]}
        # ---- e2e: BASELINE.json's second metric, the drop-in next to the unmodified reference (last: it needs the HBM) --------
        if single and os.environ.get("MEME_BENCH_E2E", "1") != "0":
            if not ref_prefix:
                out["e2e"] = {"skipped": "compiled reference / reference-format index not available on this box"}
            elif time.time() - T_START > budget - 500:
                out["e2e"] = {"skipped": "wall budget (%d s) nearly used up after %.0f s" % (budget, time.time() - T_START)}
            else:
                try:
                    del text, sa
                    ctx.close()
                    ctx = None
                    del keep, d_reads, d_off
                    torch.cuda.empty_cache()
                    if world > 1:                    # the other ranks' processes release their GPUs when they exit
                        wait_for_free_gpus(world)
                    # BASELINE configs[2] at its stated size: 10 M pairs (MEME_BENCH_E2E_PAIRS is the opt-down for probes)
                    out["e2e"] = e2e_leg(ref_prefix, fwd, int(os.environ.get("MEME_BENCH_E2E_PAIRS", "10000000")), cores, devices=world,
                                         refcache=refcache)
                except Exception as e:
                    log("e2e leg failed: %r" % (e,))
                    out["e2e"] = {"failed": repr(e)[:300]}
                # the same comparison on the repeat-dense genome (round 6): its own 128-Mbp index, so the reference's start-up is seconds, not minutes
                rd = out.get("repeat_dense")
                if isinstance(rd, dict) and rd_prefix and os.environ.get("MEME_BENCH_RD_E2E", "1") != "0":
                    try:
                        rd["e2e"] = e2e_leg(rd_prefix, rd_genome, int(os.environ.get("MEME_BENCH_RD_E2E_PAIRS", "1000000")), cores, devices=world, refcache=refcache, seed=79, slices_ok=False)
                    except Exception as e:
                        log("repeat_dense e2e failed: %r" % (e,))
                        rd["e2e"] = {"failed": repr(e)[:300]}
                c4 = out.get("config4_class")
                # configs[4]'s read class end to end: on the 128-Mbp index as well unless MEME_BENCH_C4_E2E_INDEX=bench (rounds 4-5: the GRCh38-sized one --
                # 193 s of wall, 170 of them the reference expanding that index a third time, for 18 s of measurement: VERDICT r05 item 9)
                c4_on_bench = os.environ.get("MEME_BENCH_C4_E2E_INDEX", "rd") == "bench" or not rd_prefix
                if isinstance(c4, dict) and "seeding" in c4 and os.environ.get("MEME_BENCH_C4_E2E", "1") != "0":
                    if time.time() - T_START > budget - 450:
                        c4["e2e"] = {"skipped": "wall budget (%d s) nearly used up after %.0f s" % (budget, time.time() - T_START)}
                    else:
                        try:
                            c4["e2e"] = e2e_leg(ref_prefix if c4_on_bench else rd_prefix, fwd if c4_on_bench else rd_genome, int(os.environ.get("MEME_BENCH_C4_E2E_PAIRS", "500000")), cores,
                                                devices=world, refcache=refcache, read_len=250, sub=0.05, indel=0.0075, seed=4005)
                            c4["e2e"]["index"] = "the benchmark's GRCh38-sized index" if c4_on_bench else "the repeat-dense leg's 128-Mbp index (MEME_BENCH_C4_E2E_INDEX=bench: the GRCh38-sized one, as in rounds 4-5)"
                        except Exception as e:
                            log("config4_class e2e failed: %r" % (e,))
                            c4["e2e"] = {"failed": repr(e)[:300]}
        # ---- roofline.traffic measured in this run (the committed pass stays as the fall-back, labelled as such) -----------------------
        if single and os.environ.get("MEME_BENCH_PMC", "1") != "0" and time.time() - T_START < budget - 200:
            try:
                if ctx is not None:               # (the passes build the index again: the HBM has to be free)
                    ctx.close()
                    ctx = None
                    keep = d_reads = d_off = None
                    torch.cuda.empty_cache()
                if reads_sum0 is not None and int(reads.reshape(-1).view(np.uint64).sum(dtype=np.uint64)) != reads_sum0:
                    log("the benchmark's batch on the host is no longer what was sampled: a leg changed it in place")
                # (the passes build the whole index again next to this process: everything it still holds in HBM goes first)
                d_text = d_pos5 = d_l2 = d_l1 = d_ent = d_pac = keep = d_reads = d_off = pre = None
                d_text0 = d_s = d_pos50 = d_pac0 = d_ent0 = d_l2_0 = d_l1_0 = None          # (the device-built index's tensors are names of this function too: 146 GB in call R6)
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                log("counter passes: %.0f of %.0f GB of HBM free" % tuple(x / 1e9 for x in torch.cuda.mem_get_info(local)))
                live = live_pmc_traffic(reads_file=reads_file, expect={k: out["config"][k] for k in ("smems_per_read", "hits_per_read", "searches_per_read") if k in out["config"]})
                reads_file = None
                if live:
                    out["roofline"]["traffic"] = (2 * live["fetch_kb"] + live["write_kb"]) * 1024.0
                    out["roofline"]["traffic_source"] = ("this run: bench.py re-executed under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, 2 launches of the "
                                                         "stage each, counters summed over k_seed + k_reseed*); (2 x FETCH_SIZE + WRITE_SIZE) x 1024 as the MI355X guide prescribes")
                    out["roofline"]["traffic_over_algorithmic"] = out["roofline"]["traffic"] / (bpr * nreads)
                    out["roofline"]["traffic_counters_kb_per_launch"] = {"FETCH_SIZE": live["fetch_kb"], "WRITE_SIZE": live["write_kb"]}
                    out["roofline"]["traffic_counter_passes"] = {k: live[k] for k in live if k.endswith("_kernels") or k.endswith("_pass_work")}
                    # The stage's accesses are random 128-byte lines of which a few dozen bytes are used: the memory system delivers ~50 G such lines per second
                    # whatever is used of them (scripts/microbench/gather_roofline.hip, profiles/r01_gather_roofline.md), which -- not 8 TB/s of useful bytes -- is
                    # the ceiling of this access pattern (frac 0.17-0.20 of the byte roofline).  Lines fetched per second against that ceiling:
                    lines_per_s = 2.0 * live["fetch_kb"] * 1024.0 / 128.0 / (k_ms * 1e-3)
                    out["roofline"]["random_line_ceiling_lines_per_s"] = 50e9
                    out["roofline"]["lines_fetched_per_s"] = lines_per_s
                    out["roofline"]["frac_of_random_line_ceiling"] = lines_per_s / 50e9
            except Exception as e:
                log("live pmc passes skipped: %r" % (e,))
        print(json.dumps(out), flush=True)
        if not sample_parity:
            rc_exit = 1
    if ctx is not None:
        ctx.close()
    sys.exit(rc_exit)


if __name__ == "__main__":
    main()
