#!/bin/bash
# Round 6, GPU call N3 (under twelve minutes).  The tests that the last changes touch: CIGAR kernels (16 / 32 / 64-lane classes), SAM identity (one walk over the
# records, 64-letter conversion, lower case), determinism, mate rescue, extension; then the CIGAR leg's times with the 64-lane class.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06n3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gcig.py tests/test_gpu_sam_e2e.py tests/test_gpu_determinism.py tests/test_gpu_mate.py tests/test_gpu_ext.py -q -m gpu --durations=12 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rc |^E  |Error" $O/pytest.log | tail -8 | cut -c1-300
MEME_BENCH_MBP=256 MEME_BENCH_READS=2000000 MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=1 MEME_BENCH_C4=1 MEME_BENCH_C4_E2E=0 MEME_BENCH_RD=0 \
MEME_BENCH_PMC=0 MEME_BENCH_PARITY_READS=20000 MEME_BENCH_EXT_CHECK=0 timeout 100 python bench.py --steps 2 --warmup 1 > $O/bench_groups1.json 2> $O/bench_groups1.err; echo "rc $?" >> $O/bench_groups1.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_groups1.json").read().strip().split("\n")[-1])
    for name, e in (("150bp", d["ext"]), ("250bp", d["config4_class"]["ext"])):
        c = e["cigar"]
        print(name, "kernel_ms %.2f" % c["kernel_ms"], "matches_oracle", c["matches_oracle"], c.get("jobs_by_kernel"))
except Exception as ex:
    print("no line", ex)
PY
