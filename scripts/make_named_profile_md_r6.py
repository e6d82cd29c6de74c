#!/usr/bin/env python3
"""gpurun_out/prof_named_r6/ (written by scripts/profile_named_r6.sh) -> profiles/r06_named_config.md + profiles/pmc_named.json.
The SA-search stage is several kernels since round 4 (k_seed<4> + k_reseed*): counters are summed over them.  The PMC passes run
--steps 2 --warmup 1 = 3 stage launches of 10 M reads + one 50 000-read parity slice (0.17 % of the reads: inside the sums, ignored)."""
import json, os, re, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(REPO, "gpurun_out", "prof_named_r6") + "/"
STAGE = ("k_seed<4>", "k_reseed")


def table(f):
    out = []
    for l in open(D + f).read().splitlines():
        c = [x.strip() for x in l.strip("|").split("|")]
        if len(c) >= 5 and any(k in c[0] for k in STAGE):
            out.append(c)
    return out


def counter(f, name):
    per = {}
    for c in table(f):
        if c[1] == name:
            per[c[0]] = per.get(c[0], 0.0) + float(c[4])
    return per


def trace():
    rows = {}
    for l in open(D + "trace_seed.md").read().splitlines():
        c = [x.strip() for x in l.strip("|").split("|")]
        if len(c) >= 7 and any(k in c[0] for k in STAGE + ("k_gather", "k_pack_reads", "k_build_plcp")):
            rows[c[0]] = (int(c[1]), float(c[2]), float(c[3]), float(c[4]), float(c[5]))
    return rows


traced = json.loads(open(D + "bench_traced_seed.json").read().strip().splitlines()[-1])
bench_path = os.path.join(REPO, "profiles", sys.argv[1] if len(sys.argv) > 1 else "r06_bench_default_b.json")
bench = json.loads(open(bench_path).read().strip().splitlines()[-1])
T = trace()
launches = 6                                                      # 1 warm-up + 5 timed launches of 10 M reads in the traced run (+ the parity slice)
big = {k: (v[4] - v[2] / 1000.0 * (v[0] // 7 if v[0] >= 7 else 0)) / launches for k, v in T.items()}   # total minus the slice's (minimum) dispatches
fetch, write, miss, hit = counter("pmc_fetch.md", "FETCH_SIZE"), counter("pmc_write.md", "WRITE_SIZE"), counter("pmc_write.md", "TCC_MISS_sum"), counter("pmc_write.md", "TCC_HIT_sum")
valu, salu = counter("pmc_sq.md", "SQ_INSTS_VALU"), counter("pmc_sq2.md", "SQ_INSTS_SALU")
S = lambda d: sum(d.values()) / 3.0
seed_key = [k for k in T if "k_seed<4>" in k][0]
k_seed_ms = big[seed_key]
stage_ms = sum(v for k, v in big.items() if any(s in k for s in STAGE))
alg = bench["roofline"]["algorithmic_bytes_per_read"] * 1e7
searches = bench["roofline"]["work_per_read"]["searches"] * 1e7
windows = traced["config"]["windows_per_search"] * traced["config"]["searches_per_read"] * 1e7
traffic = (2 * S(fetch) + S(write)) * 1024
pmc = json.load(open(os.path.join(REPO, "profiles", "pmc_named.json")))
pmc.update({"kernel": "k_seed<4> + k_reseed + k_reseed_emit + k_reseed_search + k_reseed_resume (the SA-search stage)",
            "fetch_size_kb_per_launch": S(fetch), "write_size_kb_per_launch": S(write), "tcc_miss_lines_per_launch": S(miss), "valu_insts_per_launch": S(valu),
            "source": "profiles/r06_named_config.md (scripts/profile_named_r6.sh: separate rocprofv3 --pmc passes of the same bench command, round 6; all kernels of the stage summed)",
            "note": "per-launch = column sums over the stage's kernels / 3 launches of 10 M reads (the passes run --steps 2 --warmup 1; a 50 000-read parity slice is inside the sums). "
                    "FETCH_SIZE tallies 128-byte line fills at 64 B on gfx950 (MI355X_MICROARCH.md); cross-check: TCC_MISS_sum x 128 B = %.0f GB vs 2 x FETCH_SIZE = %.0f GB"
                    % (S(miss) * 128 / 1e9, 2 * S(fetch) * 1024 / 1e9)})
pmc["reference_cpu"] = {k: v for k, v in bench["cpu_baseline"].items() if k != "port"}
json.dump(pmc, open(os.path.join(REPO, "profiles", "pmc_named.json"), "w"), indent=1)
md = ["# Round 6 -- named configuration (BASELINE.json configs[1]) on one MI355X: bench line, kernel trace, PMC passes\n",
      "Produced by `scripts/profile_named_r6.sh` in ONE gpurun call (ROCm 7.2, rocprofv3): `python bench.py --steps 5 --warmup 1` under `rocprofv3 --kernel-trace --stats` "
      "(seeding only: the CPU, e2e, chain, ext and bsw legs off, the in-run parity check cut to one 50 000-read slice), then four separate `--pmc` passes (`--steps 2 --warmup 1`).  "
      "The plain default run of the same code is `profiles/%s`.  Tables: `scripts/rocpd_summary.py`; this file: `scripts/make_named_profile_md_r6.py`.\n" % os.path.basename(bench_path),
      "## 1. The plain default run\n```\n" + json.dumps({k: bench[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "roofline", "cpu_baseline")}) + "\n```\n",
      "## 2. Kernel trace (bench line of the traced run: %.1f M reads/s, stage %.2f ms by HIP events)\n" % (traced["value"] / 1e6, traced["roofline"]["kernel_ms"]),
      "| kernel | dispatches | avg µs | min µs | max µs | total ms | per 10 M-read launch, ms |\n|---|---|---|---|---|---|---|"]
for k, v in sorted(T.items(), key=lambda kv: -kv[1][4]):
    md.append("| %s | %d | %.1f | %.1f | %.1f | %.2f | %s |" % (k, v[0], v[1], v[2], v[3], v[4], ("%.2f" % big[k]) if "plcp" not in k else "one-off (staging)"))
md.append("\nThe stage per 10 M-read launch: `k_seed<4>` %.2f ms + the re-seeding kernels %.2f ms = **%.2f ms** (HIP events in `bench.py`: %.2f ms in the traced run, %.2f ms in the plain run).\n"
          % (k_seed_ms, stage_ms - k_seed_ms, stage_ms, traced["roofline"]["kernel_ms"], bench["roofline"]["kernel_ms"]))
md.append("## 3. PMC passes (each its own run; sums over 3 launches of 10 M reads + the parity slice)\n")
for f, t in (("pmc_fetch.md", "FETCH_SIZE"), ("pmc_write.md", "WRITE_SIZE / TCC"), ("pmc_sq.md", "SQ, pass 1"), ("pmc_sq2.md", "SQ, pass 2")):
    md.append("### %s\n\n| kernel | counter | dispatches | mean per dispatch | sum |\n|---|---|---|---|---|" % t)
    md += ["| " + " | ".join(c) + " |" for c in table(f)]
    md.append("")
md.append("## 4. Derived, per 10 M-read launch of the stage\n\n| Quantity | Value |\n|---|---|")
md.append("| stage time | %.2f ms (trace); `k_seed<4>` alone %.2f ms |" % (stage_ms, k_seed_ms))
md.append("| algorithmic bytes | %.1f GB (%.0f B/read x 10 M; the reference's own probe sequence, 66.9 searches per read) -> %.0f GB/s = **%.3f of 8 TB/s** (round 4: 0.162) |"
          % (alg / 1e9, bench["roofline"]["algorithmic_bytes_per_read"], bench["roofline"]["achieved"], bench["roofline"]["frac"]))
md.append("| HBM traffic | (2 x FETCH_SIZE %.4g KB + WRITE_SIZE %.4g KB) x 1 024 = **%.0f GB** = %.2f x algorithmic (round 4: 234 GB, 2.9 x); cross-check TCC_MISS %.4g lines x 128 B = %.0f GB |"
          % (S(fetch), S(write), traffic / 1e9, traffic / alg, S(miss), S(miss) * 128 / 1e9))
md.append("| of which `k_seed<4>` | FETCH %.4g KB, TCC_MISS %.4g lines = %.1f G lines/s over its %.2f ms = %.0f %% of the random-line ceiling (50 G lines/s); %.2f missed lines per window search (%.3g windows per launch) |"
          % (fetch[seed_key] / 3, miss[seed_key] / 3, miss[seed_key] / 3 / k_seed_ms / 1e6, k_seed_ms, miss[seed_key] / 3 / k_seed_ms / 1e6 / 50 * 100, miss[seed_key] / 3 / windows, windows))
md.append("| VALU, `k_seed<4>` | SQ_INSTS_VALU %.3g per launch (round 4: 2.35e10); x 4 cycles / 1 024 SIMDs / (%.2f ms x 2.4 GHz) = **%.0f %% of the issue cycles**; %.1f wave-instructions per window search; SALU %.0f %% of the VALU count |"
          % (valu[seed_key] / 3, k_seed_ms, valu[seed_key] / 3 * 4 / 1024 / (k_seed_ms * 1e-3 * 2.4e9) * 100, valu[seed_key] / 3 / windows, salu[seed_key] / valu[seed_key] * 100))
md.append("| VALU, re-seeding kernels | %.3g per launch in all five: latency-bound lane-per-read code (SQ_WAIT_INST_ANY dominates their wave cycles) |" % ((S(valu) - valu[seed_key] / 3)))
md.append("| searches the stage stands for | %.3g per launch (the reference's count: the roofline numerator); window searches actually issued by `k_seed<4>`: %.3g (%.0f %%); "
          "answered from the plcp table or by the batch searches of `k_reseed_*`: the rest |" % (searches, windows, windows / searches * 100))
open(os.path.join(REPO, "profiles", "r06_named_config.md"), "w").write("\n".join(md) + "\n")
print("stage %.2f ms (k_seed %.2f), traffic %.0f GB = %.2f x algorithmic, k_seed VALU %.0f %%" % (stage_ms, k_seed_ms, traffic / 1e9, traffic / alg, valu[seed_key] / 3 * 4 / 1024 / (k_seed_ms * 1e-3 * 2.4e9) * 100))
