#!/bin/bash
# Round 5, GPU call N.  SURVEY 8 (f)2/(f)4: where the host's thread time goes in the SAM phase now that bwa_gen_cigar2, mem_aln2sam and the purged
# records have left it (bwa-meme_dropin_prof: compile-time TSC timers), 2 M pairs, same box: HEAD twice, the profiling build, HEAD at -t 32.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05n; mkdir -p $O
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_PMC=0 MEME_BENCH_E2E_PAIRS=2000000 MEME_BENCH_E2E_SKIP_REF=1 MEME_BENCH_E2E_KEEP_DIFF=1 \
MEME_BENCH_E2E_DROPIN_EXE="bwa-meme_dropin,bwa-meme_dropin_prof,bwa-meme_dropin@X=1,bwa-meme_dropin_prof@X=1" \
MEME_BENCH_E2E_STDERR=$O/e2e MEME_BENCH_PARITY_READS=50000 \
timeout 1500 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err
echo "bench rc $?" >> $O/bench.err
grep "e2e:" $O/bench.err
grep -h "meme-dropin-prof" $O/e2e/bwa-meme_dropin_prof_150bp.stderr | head -30
