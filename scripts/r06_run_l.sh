#!/bin/bash
# Round 6, GPU call L.  SURVEY 8(b) threading model / 8(d): where the bound aligner's process CPU goes, by thread role (the figure mem_process_seqs follows on a host with a CPU quota).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06l; mkdir -p $O
V="bwa-meme_dropin,bwa-meme_dropin@MEME_DROPIN_CIGAR=0,bwa-meme_dropin@MEME_DROPIN_MATESW=0,bwa-meme_dropin@MEME_DROPIN_SAM=0,bwa-meme_dropin@X=2"
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_RD=0 MEME_BENCH_PMC=0 MEME_BENCH_E2E_SKIP_REF=1 MEME_BENCH_E2E_PAIRS=4000000 MEME_BENCH_E2E_SLICES=0 \
MEME_BENCH_E2E_DROPIN_EXE="$V" MEME_BENCH_PARITY_READS=50000 MEME_BENCH_E2E_STDERR=$O/e2e timeout 1500 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
grep -E "e2e:|bench rc|failed" $O/bench.err | cut -c1-200
for f in $O/e2e/*.stderr; do echo "== $f"; grep -h "process CPU .* by thread role" $f | cut -c1-900; done
