#!/usr/bin/env python3
"""gpurun_out/prof_named_r<N>/ (written by scripts/profile_named_r<N>.sh) -> profiles/r0<N>_named_config.md + profiles/pmc_named.json.
python scripts/make_named_profile_md.py [N]   (default 3; bench.json of round 3 = the plain default run kept as profiles/r03_bench_default_d.json)
Per-launch figures: the PMC passes run --steps 2 --warmup 1 = 3 launches of 10 M reads + the 20 k-read parity sample."""
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = int(sys.argv[1]) if len(sys.argv) > 1 else 3
D = os.path.join(REPO, "gpurun_out", "prof_named_r%d" % RND) + "/"


def rows(f, pat):
    return [l for l in open(D + f).read().splitlines() if re.search(pat, l)]


def total(f, counter, kernel="k_seed<4>"):
    for l in rows(f, re.escape(kernel)):
        c = [x.strip() for x in l.strip("|").split("|")]
        if len(c) >= 5 and c[1] == counter:
            return float(c[4])
    raise SystemExit("counter %s not found in %s" % (counter, f))


def trace_row(kernel="k_seed<4>"):
    for l in rows("trace_seed.md" if os.path.exists(D + "trace_seed.md") else "trace.md", re.escape(kernel)):
        c = [x.strip() for x in l.strip("|").split("|")]
        return int(c[1]), float(c[2]), float(c[3]), float(c[4]), float(c[5])      # calls avg min max total_ms
    raise SystemExit("kernel not in trace")


bench_path = D + "bench.json" if os.path.exists(D + "bench.json") else os.path.join(REPO, "profiles", "r%02d_bench_default_a.json" % RND)
bench = json.loads(open(bench_path).read().strip().splitlines()[-1])
traced = json.loads(open(D + "bench_traced.json").read().strip().splitlines()[-1])
fetch = total("pmc_fetch.md", "FETCH_SIZE") / 3
write = total("pmc_write.md", "WRITE_SIZE") / 3
miss = total("pmc_write.md", "TCC_MISS_sum") / 3
hit = total("pmc_write.md", "TCC_HIT_sum") / 3
valu = total("pmc_sq.md", "SQ_INSTS_VALU") / 3
salu = total("pmc_sq2.md", "SQ_INSTS_SALU") / 3
calls, avg, mn, mx, tot = trace_row()
k_avg = (tot - mn / 1000.0) / (calls - 1)
searches = bench["roofline"]["work_per_read"]["searches"] * 1e7
pmc = json.load(open(os.path.join(REPO, "profiles", "pmc_named.json")))
pmc.update({"fetch_size_kb_per_launch": fetch, "write_size_kb_per_launch": write, "tcc_miss_lines_per_launch": miss, "valu_insts_per_launch": valu,
            "source": "profiles/r%02d_named_config.md (scripts/profile_named_r%d.sh: separate rocprofv3 --pmc passes of the same bench command, round %d)" % (RND, RND, RND),
            "note": "per-launch = column sum / 3 timed launches of 10 M reads (the passes run --steps 2 --warmup 1; the fourth dispatch is the 20 k-read parity sample). "
                    "FETCH_SIZE tallies 128-byte line fills at 64 B on gfx950 (MI355X_MICROARCH.md); cross-check: TCC_MISS_sum x 128 B = %.0f GB vs 2 x FETCH_SIZE = %.0f GB"
                    % (miss * 128 / 1e9, 2 * fetch * 1024 / 1e9)})
pmc["reference_cpu"] = {k: v for k, v in bench["cpu_baseline"].items() if k != "port"}
json.dump(pmc, open(os.path.join(REPO, "profiles", "pmc_named.json"), "w"), indent=1)
alg = bench["roofline"]["algorithmic_bytes_per_read"] * 1e7
md = []
md.append("# Round %d -- named configuration" % RND + " (BASELINE.json configs[1]) on one MI355X: bench line, kernel trace, PMC passes\n")
md.append("Produced by `scripts/profile_named_r%d.sh`" % RND + " in ONE gpurun call (ROCm 7.2, rocprofv3): a plain `python bench.py --steps 5 --warmup 1`, the same command under "
          "`rocprofv3 --kernel-trace --stats` (without the two CPU legs), then four separate `--pmc` passes (`--steps 2 --warmup 1`, seeding only).  Tables: "
          "`scripts/rocpd_summary.py` over the rocpd databases; this file: `scripts/make_named_profile_md.py`.\n")
md.append("## 1. The plain run (`bench.json`)\n```\n" + json.dumps(bench) + "\n```\n")
if os.path.exists(D + "bench.err"):
    md.append("stderr of that run:\n```\n" + "\n".join(l for l in open(D + "bench.err").read().splitlines() if "amdgpu.ids" not in l) + "\n```\n")
md.append("## 2. Kernel trace of the same command (`rocprofv3 --kernel-trace --stats`; bench line of the traced run: %.1f M reads/s, k_seed %.2f ms by HIP events)\n"
          % (traced["value"] / 1e6, traced["roofline"]["kernel_ms"]))
md.append("\n".join(open(D + "trace.md").read().splitlines()[1:40]) + "\n")
md.append(("In the seeding-only trace of the same command (`trace_seed`: the chain / ext legs of the full trace seed 2 M-read batches of their own, which would mix into the average): "
           if os.path.exists(D + "trace_seed.md") else "") +
          "`k_seed<4>`: %d dispatches = %d launches of 10 M reads (1 warm-up + 5 timed) + the 20 000-read parity sample (%.2f ms): **%.2f ms per 10 M-read launch**, max %.2f ms; "
          "`bench.py`'s HIP-event figure for the timed region: %.2f ms (it includes any overflow-tier launch).\n" % (calls, calls - 1, mn / 1000, k_avg, mx / 1000, traced["roofline"]["kernel_ms"]))
md.append("## 3. PMC passes (each its own run; 3 launches of 10 M reads + the parity sample per pass)\n")
for f, t in (("pmc_fetch.md", "FETCH_SIZE"), ("pmc_write.md", "WRITE_SIZE / TCC"), ("pmc_sq.md", "SQ, pass 1"), ("pmc_sq2.md", "SQ, pass 2")):
    md.append("### %s\n" % t)
    md.append("| kernel | counter | dispatches | mean per dispatch | sum |\n|---|---|---|---|---|")
    md.append("\n".join(r for r in rows(f, r"k_seed<4>|k_gather") if re.search(r"\| (FETCH|WRITE|TCC|SQ)_", r)) + "\n")
md.append("## 4. Derived, per 10 M-read launch of `k_seed<4>`\n")
md.append("| Quantity | Value |\n|---|---|")
md.append("| kernel time | %.2f ms (trace), %.2f ms (HIP events of the plain run) |" % (k_avg, bench["roofline"]["kernel_ms"]))
md.append("| algorithmic bytes | %.1f GB (%.0f B/read x 10 M) -> %.0f GB/s = %.3f of 8 TB/s |" % (alg / 1e9, bench["roofline"]["algorithmic_bytes_per_read"], bench["roofline"]["achieved"], bench["roofline"]["frac"]))
md.append("| HBM traffic | FETCH_SIZE %.4g KB x 2 (gfx950 tallies a 128-byte fill as 64 B) + WRITE_SIZE %.4g KB = **%.0f GB** = %.1f x algorithmic; cross-check TCC_MISS %.4g lines x 128 B = %.0f GB |"
          % (fetch, write, (2 * fetch + write) * 1024 / 1e9, (2 * fetch + write) * 1024 / alg, miss, miss * 128 / 1e9))
md.append("| L2-miss lines | %.3g per launch = %.1f G lines/s = %.0f %% of the random-line ceiling (50 G lines/s, profiles/r01_gather_roofline.md); %.2f per search |"
          % (miss, miss / k_avg / 1e6, miss / k_avg / 1e6 / 50 * 100, miss / searches))
md.append("| VALU | SQ_INSTS_VALU %.3g per launch; x 4 cycles / 1 024 SIMDs / (%.2f ms x 2.4 GHz) = **%.0f %% of the issue cycles**; %.1f wave-instructions per search |"
          % (valu, k_avg, valu * 4 / 1024 / (k_avg * 1e-3 * 2.4e9) * 100, valu / searches))
md.append("| SALU | SQ_INSTS_SALU %.3g per launch (%.0f %% of the VALU count) |" % (salu, salu / valu * 100))
md.append("| L2 hits | TCC_HIT %.3g per launch: the kernel's working set does not live in L2 (model 8.6 GB, entries 99 GB) |" % hit)
if RND == 2:
    md.append("\nSame kernel as round 1 except the 32-byte model records, the bounded partial-layer index and 128 first-pass SMEM slots per read; the one-lane-per-read kernel that was "
              "built to halve the lines is documented in `r02_seed_v2_experiment.md`.\n")
else:
    md.append("\n`k_seed` is the round-2 kernel (why the reference's ISA shortcut was not added: DESIGN.md section 8).  New in this trace: the chaining tiers (`k_chain_route`, `k_chain`, "
              "`k_chain_lds`, `k_chain_wave`), the extension stage (`k_ext_*` around `k_bsw_lane`) and the CIGAR kernel (`k_gcig`) and the mate-rescue kernel (`k_kswv`) of bench.py's `chain` / `ext` / `kswv` legs.\n")
    md.append("## 5. The kernels of the stages behind seeding in the same trace (2 M reads per leg)\n")
    md.append("| kernel | calls | avg_us | min_us | max_us | total_ms | pct |\n|---|---|---|---|---|---|---|")
    md.append("\n".join(r for r in open(D + "trace.md").read().splitlines() if re.search(r"k_chain|k_ext|k_gcig|k_kswv|k_bsw|k_scan|k_gather|k_pack_reads|k_offsets", r)) + "\n")
open(os.path.join(REPO, "profiles", "r%02d_named_config.md" % RND), "w").write("\n".join(md))
print("k_seed %.2f ms/launch, traffic %.0f GB, miss lines/search %.2f, VALU %.0f %%" % (k_avg, (2 * fetch + write) * 1024 / 1e9, miss / searches, valu * 4 / 1024 / (k_avg * 1e-3 * 2.4e9) * 100))
