#!/bin/bash
# Round 6, GPU call N5 (under thirteen minutes).  The three test files call N4 did not reach (its 540 s ran out), then the host-CPU accounting of the bound aligner after
# the two cuts (64-letter conversion, one walk over the records): call L2's run again -- 4 M pairs, GRCh38-sized index, helper team's CPU by host loop.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06n5; mkdir -p $O
timeout 420 python -m pytest tests/test_gpu_scale.py tests/test_gpu_sam_scale.py tests/test_gpu_bench_multirank.py -q -m gpu --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rc |^E  |Error|s call" $O/pytest.log | tail -12 | cut -c1-300
V="bwa-meme_dropin,bwa-meme_dropin@X=2"
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_RD=0 MEME_BENCH_PMC=0 MEME_BENCH_E2E_SKIP_REF=1 MEME_BENCH_E2E_PAIRS=4000000 MEME_BENCH_E2E_SLICES=0 \
MEME_BENCH_E2E_DROPIN_EXE="$V" MEME_BENCH_PARITY_READS=50000 MEME_BENCH_E2E_STDERR=$O/e2e timeout 330 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
grep -E "e2e:|bench rc|failed" $O/bench.err | cut -c1-200
for f in $O/e2e/*.stderr; do echo "== $f"; grep -h "by thread role\|by host loop" $f | cut -c1-1400; done
