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
is built on the host (about 4-6 minutes on the 256-thread box) and cached in /dev/shm for the next invocation.

Workload knobs (environment): MEME_BENCH_MBP (genome size in Mbp, default 3100), MEME_BENCH_READS (reads per GPU
per step, default 10,000,000), MEME_BENCH_BITS (P-RMI leaves = 2^bits, default: the reference's rule),
MEME_BENCH_WAVES (resident wavefronts per CU of the SA-search kernel), MEME_BENCH_CPU (reference | port | 0; default: the compiled
reference up to 1 Gbp, the restated port above), MEME_BENCH_CPU_READS (sample size), MEME_BENCH_CACHE (0 disables).
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
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pymeme import hipapi, hostapi, synth, workload  # noqa: E402

READ_LEN = 150


def log(msg):
    print("[bench r%s] %s" % (os.environ.get("RANK", "0"), msg), file=sys.stderr, flush=True)


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


def cpu_baseline_reference(fwd, text, sa, l1, l2, reads, cores):
    """Times the COMPILED REFERENCE (oracle/_ref/learned_seeding_mode3 = test/Learned_seeding_big_read.cpp,
    MODE=3, AVX-512 build) on the same index and a sample of the same reads.  Measurement only."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_py as O
    exe = os.path.join(REPO, "oracle", "_ref", "learned_seeding_mode3")
    big = text.nbytes > (1 << 31)
    tmp = tempfile.mkdtemp(prefix="meme_cpu_", dir="/dev/shm" if big and os.path.isdir("/dev/shm") else None)
    try:
        prefix = os.path.join(tmp, "ref.fa")
        t0 = time.time()
        hostapi.write_index(prefix, fwd, text, sa, l1, l2, n_contigs=4)
        fq = os.path.join(tmp, "sample.fq")
        fastq_bytes(reads).tofile(fq)
        log("cpu_baseline: index + FASTQ written in %.1f s" % (time.time() - t0))
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
                          "(MODE=3, AVX-512 build, steps=3) timed by its own rdtsc counter" % reads.shape[0]}
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


def algorithmic_bytes_per_read(text, sa, l1, l2, reads):
    """SURVEY 8(d) per-read figure from the instrumented restatement, on a sample of the same reads."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_py as O
    idx = O.Index(text, sa)
    n = reads.shape[0]
    off = np.arange(0, (n + 1) * READ_LEN, READ_LEN, dtype=np.int64)
    _, _, ctr = O.refpath_seed_batch(idx, l1, l2, reads, off, threads=0, keep_smems=False)
    return O.algorithmic_bytes(ctr, n * READ_LEN) / n, {k: v / n for k, v in ctr.items()}


def bsw_leg(ctx, dev, world):
    """Second kernel of the path: banded seed extension on rank 0's GPU, pairs resident in HBM.
    Reported beside the headline metric: pairs/s, GCUPS (DP cells from the CPU restatement's counter) and the
    CPU restatement on all host cores for the same pairs."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_py
    npairs = int(os.environ.get("MEME_BENCH_BSW_PAIRS", "2000000"))
    pairs, ref, qer, base = workload.make_bsw_pairs(npairs, seed=77, read_len=READ_LEN)
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
    got = d_pairs.cpu().numpy().view(hipapi.SEQPAIR)[:base]
    # cells and the CPU figure from the restatement of scalarBandedSWA (the oracle): checker + baseline only
    chk = pairs[:base].copy().view(oracle_py.SEQPAIR_DTYPE)
    cells = oracle_py.bsw_batch(chk, ref, qer, 100, threads=0)
    same = all(np.array_equal(got[f], chk[f]) for f in ("score", "tle", "gtle", "qle", "gscore", "max_off"))
    reps = npairs / base
    cores = os.cpu_count() or 1
    big = np.tile(pairs[:base], max(1, int(400000 // base)))
    big_chk = big.copy().view(oracle_py.SEQPAIR_DTYPE)
    t0 = time.perf_counter()
    oracle_py.bsw_batch(big_chk, ref, qer, 100, threads=cores)
    cpu_dt = time.perf_counter() - t0
    return {"metric": "bsw_pairs_per_sec", "value": npairs / (k_ms * 1e-3), "unit": "pairs/s", "per": "gpu",
            "pairs": npairs, "band_w": 100, "kernel_ms": k_ms, "cells_per_pair": cells / base,
            "gcups": cells * reps / (k_ms * 1e-3) / 1e9, "matches_oracle": bool(same),
            # integer-VALU roofline of the lane-per-pair kernel: 35 VALU instructions per DP cell in its inner loop (ISA count),
            # against 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz int32 lane-ops/s
            "roofline": {"bound": "valu-int32", "ops_per_cell": 35, "peak": 39.3, "unit": "Tops/s",
                         "achieved": 35 * cells * reps / (k_ms * 1e-3) / 1e12,
                         "frac": 35 * cells * reps / (k_ms * 1e-3) / 1e12 / 39.3},
            "cpu_baseline": {"value": big.shape[0] / cpu_dt, "unit": "pairs/s", "cores": cores, "kind": "port",
                             "sample": "%d pairs, scalar restatement of scalarBandedSWA on %d threads" % (big.shape[0], cores)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch multi-GPU runs through torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP backend has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # rank 0 builds the index on its host for several minutes while the others wait at the first broadcast
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=90))

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
        gpu_free = torch.cuda.mem_get_info(local)[0]
        while mbp > 64 and 2 * mbp * 1e6 * 30 + nreads * 2200 > 0.9 * gpu_free:
            mbp /= 2
            log("HBM too small for the configured genome: falling back to %.0f Mbp" % mbp)
    l_pac_t = torch.tensor([int(mbp * 1e6) & ~1], dtype=torch.int64, device=dev)
    if world > 1:
        dist.broadcast(l_pac_t, 0)
    l_pac = int(l_pac_t[0])
    n = 2 * l_pac

    # ---- index: built on rank 0's host, staged on every GPU ----------------------------------------
    meta = torch.zeros(4, dtype=torch.int64, device=dev)
    fwd = text = sa = l1 = l2 = None
    if rank == 0:
        t0 = time.time()
        fwd = synth.make_genome(l_pac, seed=11)
        cache = None
        if os.environ.get("MEME_BENCH_CACHE", "1") != "0" and os.path.isdir("/dev/shm"):
            cache = "/dev/shm/meme_bench_idx_%d_b%d" % (l_pac, bits)
        if cache and os.path.exists(cache + ".ok"):
            text = np.fromfile(cache + ".text", dtype=np.uint8)
            sa = np.fromfile(cache + ".sa", dtype=np.uint64)
            l1 = np.fromfile(cache + ".l1", dtype=hostapi.RMI_DTYPE)
            l2 = np.fromfile(cache + ".l2", dtype=hostapi.RMI_DTYPE)
            log("genome %.0f Mbp: index loaded from the /dev/shm cache in %.1f s" % (l_pac / 1e6, time.time() - t0))
        else:
            text, sa = hostapi.build_sa(fwd)
            l1, l2 = hostapi.train_prmi(text, sa, bits=bits)
            log("genome %.0f Mbp: suffix array + P-RMI (2^%d leaves, %d partial) built on host in %.1f s"
                % (l_pac / 1e6, int(np.log2(l2.shape[0])), l1.shape[0], time.time() - t0))
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
    ctx = hipapi.Context(local)
    if os.environ.get("MEME_BENCH_WAVES"):
        ctx.set_tuning("seed_waves_per_cu", int(os.environ["MEME_BENCH_WAVES"]))
    L = hipapi.lib()
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
    if world > 1:
        # one-off RCCL broadcast of the raw index images over xGMI (5 B per suffix + 1 B per base + the model tables);
        # no collective in steady state
        for t in (d_text, d_pos5, d_l2, d_l1):
            dist.broadcast(t, 0)
    torch.cuda.synchronize()
    keep = hipapi.stage_index_torch(ctx, n, d_text, d_pos5, d_l2, n_l2, d_l1, n_l1)
    del d_text, d_l2, d_l1
    torch.cuda.empty_cache()
    log("index staged in HBM in %.1f s (%.2f GB keys + %.2f GB positions)" % (time.time() - t0, 8 * n / 1e9, 5 * n / 1e9))

    # ---- reads: every rank samples its own batch ------------------------------------------------------
    genome_t = torch.empty(l_pac, dtype=torch.uint8, device=dev)
    if rank == 0:
        genome_t.copy_(torch.from_numpy(fwd))
    if world > 1:
        dist.broadcast(genome_t, 0)
    genome = fwd if rank == 0 else genome_t.cpu().numpy()
    del genome_t
    t0 = time.time()
    reads = workload.make_reads_fast(genome, nreads, READ_LEN, seed=1000 + rank)
    d_reads = torch.from_numpy(reads.reshape(-1)).to(dev)
    d_off = torch.arange(0, (nreads + 1) * READ_LEN, READ_LEN, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    log("%d reads sampled and uploaded in %.1f s" % (nreads, time.time() - t0))
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
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = step()
        tm = ctx.timings()
        kernel_ms.append((tm.seed_kernel_ms, tm.seed_gather_ms + tm.seed_pack_ms))
        windows = tm.seed_windows
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt[0])

    if rank == 0:
        k_ms = float(np.mean([k for k, _ in kernel_ms]))
        g_ms = float(np.mean([g for _, g in kernel_ms]))
        sample = reads[:20000]
        bpr, per_read = algorithmic_bytes_per_read(text, sa, l1, l2, sample)
        # parity at the benchmark's own size (index of n suffixes): the GPU seeds the same sample on its own and must
        # emit exactly as many SMEMs and hits as the CPU restatement counted (the restatement is only the checker here)
        ns = sample.shape[0]
        d_s = torch.from_numpy(sample.reshape(-1)).to(dev)
        d_so = torch.arange(0, (ns + 1) * READ_LEN, READ_LEN, dtype=torch.int64, device=dev)
        rs = ctx.seed_batch_device(d_s.data_ptr(), d_so.data_ptr(), ns, ns * READ_LEN, opt)
        sample_parity = (rs.total_smems == int(round(per_read["smems"] * ns)) and
                         rs.total_hits == int(round(per_read["hits"] * ns)))
        if not sample_parity:
            log("PARITY MISMATCH on the %d-read sample: GPU %d SMEMs / %d hits, restatement %d / %d"
                % (ns, rs.total_smems, rs.total_hits, round(per_read["smems"] * ns), round(per_read["hits"] * ns)))
        achieved = bpr * nreads / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "seeding_reads_per_sec", "value": world * nreads * a.steps / dt, "unit": "reads/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "learned-index seeding (rounds 1-3 + hit gather), %d bp SE reads vs %.0f Mbp synthetic "
                                   "genome (fwd+rc suffix array of %d entries), index and reads resident in HBM"
                                   % (READ_LEN, l_pac / 1e6, n),
                       "genome_bp": l_pac, "sa_entries": n, "reads_per_gpu_per_step": nreads, "read_len": READ_LEN,
                       "rmi_leaves_log2": int(np.log2(n_l2)), "sharding": "reads/%d ranks, index replicated by RCCL broadcast" % world,
                       "smems_per_read": res.total_smems / nreads, "hits_per_read": res.total_hits / nreads,
                       "searches_per_read": res.searches / nreads,
                       "windows_per_search": windows / max(res.searches, 1),
                       "sample_parity_with_cpu_restatement": bool(sample_parity)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": None, "kernel": "k_seed", "kernel_ms": k_ms, "gather_ms": g_ms,
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
                out["roofline"]["traffic_source"] = pmc["source"]
                out["roofline"]["l2_miss_lines_per_s"] = pmc["tcc_miss_lines_per_launch"] / (k_ms * 1e-3)
                out["roofline"]["random_line_roofline_lines_per_s"] = 50e9     # scripts/microbench/gather_roofline.hip
            except Exception as e:  # never lose the headline line over an annotation
                log("pmc annotation skipped: %r" % (e,))
                pmc = None
        cpu = None
        cpu_mode = os.environ.get("MEME_BENCH_CPU", "reference" if l_pac <= 1_000_000_000 else "port")
        if world == 1 and cpu_mode != "0":
            cores = os.cpu_count() or 1
            ns = min(nreads, int(os.environ.get("MEME_BENCH_CPU_READS", "2000000" if cpu_mode != "port" else "400000")))
            try:
                if cpu_mode != "port" and os.path.exists(os.path.join(REPO, "oracle", "_ref", "learned_seeding_mode3")) and \
                        "avx512bw" in open("/proc/cpuinfo").read():
                    cpu = cpu_baseline_reference(fwd, text, sa, l1, l2, reads[:ns], cores)
                else:
                    cpu = cpu_baseline_port(text, sa, l1, l2, reads[:min(ns, 400000)], cores)
            except Exception as e:  # the baseline is a reported extra, never the measured value
                log("cpu_baseline leg failed: %r -- falling back to the port" % (e,))
                cpu = cpu_baseline_port(text, sa, l1, l2, reads[:min(ns, 400000)], cores)
        if cpu is not None and cpu.get("kind") == "port" and pmc:
            cpu["reference_on_this_configuration"] = pmc["reference_cpu"]   # measured once (3 min of index load): see its source
        out["cpu_baseline"] = cpu
        if os.environ.get("MEME_BENCH_BSW", "1") != "0":
            try:
                out["bsw"] = bsw_leg(ctx, dev, world)
            except Exception as e:  # a secondary measurement: never lose the headline line over it
                log("bsw leg failed: %r" % (e,))
                out["bsw"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
