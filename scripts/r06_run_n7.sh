#!/bin/bash
# Round 6, GPU call N7 (short).  k_gcig_grp with the trimmed row body (two single-lane stores per row replaced by arithmetic, unconditional loads): fixtures + oracle,
# times; bench.py's trimmed paths on a probe genome: the configs[4]-class extension leg checked on a subset of its reads, the counter passes on the parent's reads.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06n7; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_gcig.py tests/test_gpu_sam_e2e.py -x -q -m gpu -k "gcig or cigar or 250bp or lower_case or identical_to_reference" > $O/pytest_gcig.log 2>&1; echo "pytest rc $?" >> $O/pytest_gcig.log; tail -3 $O/pytest_gcig.log
MEME_BENCH_MBP=256 MEME_BENCH_READS=2000000 MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=1 MEME_BENCH_C4=1 MEME_BENCH_C4_E2E=0 MEME_BENCH_RD=0 \
MEME_BENCH_PMC=0 MEME_BENCH_PARITY_READS=20000 MEME_BENCH_C4_READS=400000 MEME_BENCH_C4_EXT_CHECK_READS=100000 timeout 400 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "rc $?" >> $O/bench.err
grep -E "^\[bench|rc " $O/bench.err | cut -c1-200
python - <<PY
import json
try:
    d = json.loads(open("$O/bench.json").read().strip().split("\n")[-1])
    for name, e in (("150bp", d["ext"]), ("250bp", d["config4_class"]["ext"])):
        c = e["cigar"]
        print(name, "ext matches_oracle", e["matches_oracle"], "checked_reads", e.get("checked_reads"), "of", e["reads"], "in_rounds ok", e["in_rounds"].get("equals_the_checked_records_minus_the_purged_ones"),
              "| cigar kernel_ms %.2f" % c["kernel_ms"], c["matches_oracle"])
    print("traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_source", "")[:40])
except Exception as ex:
    print("no line", ex)
PY
