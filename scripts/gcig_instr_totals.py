#!/usr/bin/env python3
"""Instruction totals of k_gcig per launch from the counter summaries under profiles/r05_gcig_counters/ (the table in profiles/r05_gcig.md):
VALU + SALU + LDS + VMEM per launch and per SIMD, cycles per instruction per SIMD at an ASSUMED 2.4 GHz, for both read classes with the vector and the
scalar traceback walk.  SQ_ACTIVE_INST_MISC was collected in one of the four cases only; it is printed as a share, not folded into the totals."""
import os, re
D = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_gcig_counters")
def ctr(*names):
    d = {}
    for n in names:
        for line in open(os.path.join(D, n)):
            m = re.match(r"\|\s*\(anonymous namespace\)::k_gcig\(\(anonymous\s*\|\s*(\w+)\s*\|\s*\d+\s*\|\s*([0-9.e+]+)\s*\|", line)
            if m: d[m.group(1)] = float(m.group(2))
            m = re.match(r"\|\s*\(anonymous namespace\)::k_gcig\(\(anonymous namespace\)::GcigArgs\)\s*\|\s*\d+\s*\|\s*([0-9.]+)\s*\|", line)
            if m: d.setdefault("avg_us", float(m.group(1)))
    return d
cases = [("250 bp, vector walk", ctr("before_250bp_pass1.md", "before_250bp_pass2.md")),
         ("250 bp, scalar walk", ctr("scalar_walk_250bp_pass1.md", "scalar_walk_250bp_fetch_scalar.md")),
         ("150 bp, vector walk", ctr("before_150bp_pass1.md", "before_150bp_pass2.md")),
         ("150 bp, scalar walk", ctr("scalar_walk_150bp_pass1.md"))]
vmem = {"250": cases[0][1]["SQ_INSTS_VMEM_WR"] + cases[0][1]["SQ_INSTS_VMEM_RD"], "150": cases[2][1]["SQ_INSTS_VMEM_WR"] + cases[2][1]["SQ_INSTS_VMEM_RD"]}
tot = {}
for name, d in cases:
    t = d["SQ_INSTS_VALU"] + d["SQ_INSTS_SALU"] + d["SQ_INSTS_LDS"] + vmem[name[:3]]
    tot[name] = t
    cyc = d["avg_us"] * 1e-6 * 2.4e9
    print("| %s | %.3e | %.3e | %.3e | %.3e | %.3e (%.2f ms) | %.2f |" % (name, d["SQ_INSTS_VALU"], d["SQ_INSTS_SALU"], t, t / 1024, cyc, d["avg_us"] * 1e-3, cyc / (t / 1024)))
print("total changed by the scalar walk: 250 bp %+.1f %%, 150 bp %+.1f %%" % (100 * (tot["250 bp, scalar walk"] / tot["250 bp, vector walk"] - 1), 100 * (tot["150 bp, scalar walk"] / tot["150 bp, vector walk"] - 1)))
print("SQ_ACTIVE_INST_MISC (250 bp, scalar walk only): %.3e = %.1f %% on top of that case's total" % (cases[1][1]["SQ_ACTIVE_INST_MISC"], 100 * cases[1][1]["SQ_ACTIVE_INST_MISC"] / tot["250 bp, scalar walk"]))
