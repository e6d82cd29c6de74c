#!/bin/bash
# Round 6, GPU call L2.  As call L, with the helper team's CPU split by host loop.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06l2; mkdir -p $O
V="bwa-meme_dropin,bwa-meme_dropin@X=2"
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_RD=0 MEME_BENCH_PMC=0 MEME_BENCH_E2E_SKIP_REF=1 MEME_BENCH_E2E_PAIRS=4000000 MEME_BENCH_E2E_SLICES=0 \
MEME_BENCH_E2E_DROPIN_EXE="$V" MEME_BENCH_PARITY_READS=50000 MEME_BENCH_E2E_STDERR=$O/e2e timeout 1500 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
grep -E "e2e:|bench rc|failed" $O/bench.err | cut -c1-200
for f in $O/e2e/*.stderr; do echo "== $f"; grep -h "by thread role\|by host loop" $f | cut -c1-1200; done
