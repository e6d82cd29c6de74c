#!/bin/bash
# Round 6, GPU call N2 (kept under ten minutes).  SURVEY 8(f)2, the CIGAR stage: k_gcig_grp (4 / 2 jobs per wavefront) against one wavefront per job, both read classes,
# 400 k calls each on a 256-Mbp probe genome (kernel times as on the benchmark index, profiles/r05_gcig.md); then the lower-case SAM test.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06n2; mkdir -p $O
for gr in 1 0; do
MEME_TUNING="gcig_groups=$gr" MEME_BENCH_MBP=256 MEME_BENCH_READS=2000000 MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=1 MEME_BENCH_C4=1 MEME_BENCH_C4_E2E=0 MEME_BENCH_RD=0 \
MEME_BENCH_PMC=0 MEME_BENCH_PARITY_READS=20000 MEME_BENCH_EXT_CHECK=0 timeout 200 python bench.py --steps 2 --warmup 1 > $O/bench_groups$gr.json 2> $O/bench_groups$gr.err; echo "rc $?" >> $O/bench_groups$gr.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_groups$gr.json").read().strip().split("\n")[-1])
    for name, e in (("150bp", d["ext"]), ("250bp", d["config4_class"]["ext"])):
        c = e["cigar"]
        print("gcig_groups=$gr", name, "kernel_ms %.2f" % c["kernel_ms"], "matches_oracle", c["matches_oracle"], c.get("jobs_by_kernel"))
except Exception as ex:
    print("gcig_groups=$gr: no line", ex)
PY
done
timeout 200 python -m pytest tests/test_gpu_sam_e2e.py -x -q -m gpu -k "lower_case" > $O/pytest_sam.log 2>&1; echo "pytest rc $?" >> $O/pytest_sam.log; tail -3 $O/pytest_sam.log
