#!/bin/bash
# Round 6, GPU call J.  SURVEY 8(b)/(f)2: the SAM phase in two halves (the second half's pre-passes beside the first half's worker_sam) -- every SAM test, the mate-rescue tests,
# and a same-box A/B at the named configuration (halves on / off, device / host posing).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06j; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_mate.py tests/test_gpu_sam_e2e.py tests/test_gpu_determinism.py tests/test_gpu_repeat_dense.py tests/test_gpu_sam_scale.py -q -m gpu -s > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rc |repeat-dense:|stale read|Memory access|MATE_CHECK|^E  " $O/pytest.log | tail -10 | cut -c1-300
V="bwa-meme_dropin,bwa-meme_dropin@MEME_DROPIN_HALVES=0,bwa-meme_dropin@X=2,bwa-meme_dropin@MEME_DROPIN_HALVES=0@X=2,bwa-meme_dropin@X=3,bwa-meme_dropin@MEME_DROPIN_HALVES=0@MEME_DROPIN_MATE_POSE=0"
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_RD=0 MEME_BENCH_PMC=0 MEME_BENCH_E2E_SKIP_REF=1 MEME_BENCH_E2E_PAIRS=4000000 MEME_BENCH_E2E_SLICES=0 \
MEME_BENCH_E2E_DROPIN_EXE="$V" MEME_BENCH_PARITY_READS=50000 MEME_BENCH_E2E_STDERR=$O/e2e timeout 1500 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
grep -E "e2e:|bench rc|failed" $O/bench.err | cut -c1-200
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06j/bench.json").read().strip().split("\n")[-1])
e = d["e2e"]
print("extra runs identical to the first:", e["extra_runs_sam_identical_to_the_first"])
PY
ls $O/e2e | head; rm -f $O/e2e/*X=*.stderr
