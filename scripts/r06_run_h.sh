#!/bin/bash
# Round 6, GPU call H.  SURVEY 8(f)1 (extension stage: eight lanes per light read -- parity both ways, timing) + 8(f)2 (mate rescue on the device: the tests call G did not reach).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06h; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_ext.py tests/test_gpu_mate.py tests/test_gpu_repeat_dense.py tests/test_gpu_sam_e2e.py tests/test_gpu_determinism.py -q -m gpu -s > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rc |repeat-dense:|stale read|Error|^E  |MATE_CHECK" $O/pytest.log | tail -14 | cut -c1-300
for sp in 1 0; do
MEME_TUNING="ext_split=$sp" MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_C4=0 MEME_BENCH_RD=0 MEME_BENCH_PMC=0 MEME_BENCH_E2E=0 MEME_BENCH_EXT_CHECK=0 MEME_BENCH_MBP=512 MEME_BENCH_READS=2000000 \
MEME_BENCH_PARITY_READS=20000 timeout 900 python bench.py --steps 2 --warmup 1 > $O/bench_split$sp.json 2> $O/bench_split$sp.err
python - <<PY
import json
d = json.loads(open("$O/bench_split$sp.json").read().strip().split("\n")[-1])
e = d["ext"]
print("ext_split=$sp:", {k: round(e["in_rounds"][k], 2) for k in ("chain_ms", "ext_ms", "bsw_ms")}, "all at once:", {k: round(e["all_seeds_at_once"][k], 2) for k in ("ext_ms", "bsw_ms")})
PY
done
