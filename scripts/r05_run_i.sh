#!/bin/bash
# Round 5, GPU call I.  SURVEY 8 (f)2/(f)4: the CIGAR pre-pass beside mem_pestat -- SAM identity, A/B on one box.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05i; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sam_e2e.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_PMC=0 MEME_BENCH_E2E_PAIRS=4000000 MEME_BENCH_E2E_SKIP_REF=1 MEME_BENCH_E2E_KEEP_DIFF=1 \
MEME_BENCH_E2E_DROPIN_EXE="bwa-meme_dropin,bwa-meme_dropin@MEME_DROPIN_PREPASS_EARLY=0,bwa-meme_dropin@X=1,bwa-meme_dropin@MEME_DROPIN_PREPASS_EARLY=0@X=1,r04/bwa-meme_dropin_r04" \
MEME_BENCH_E2E_STDERR=$O/e2e MEME_BENCH_PARITY_READS=50000 \
timeout 1500 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err
echo "bench rc $?" >> $O/bench.err
grep -E "e2e:" $O/bench.err | tail -24
