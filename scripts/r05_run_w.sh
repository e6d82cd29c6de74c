#!/bin/bash
# Round 5, GPU call W.  SURVEY 8 (b)/(f): is the bound aligner's output reproducible?  Twelve identical runs (4 M pairs of 150-bp reads, GRCh38-sized index,
# one box), every run's SAM compared record by record with the first; differing records are kept (MEME_BENCH_E2E_KEEP_DIFF=1).  Follows up the one differing
# md5 of call B (1 of 36 runs, never reproduced) on the round's last code: extension in rounds, the CIGAR and SAM-text kernels as they stand.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05w; mkdir -p $O
V="bwa-meme_dropin"; for k in 1 2 3 4 5 6 7 8 9 10 11; do V="$V,bwa-meme_dropin@X=$k"; done
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_PMC=0 MEME_BENCH_E2E_PAIRS=4000000 MEME_BENCH_E2E_SKIP_REF=1 MEME_BENCH_E2E_KEEP_DIFF=1 \
MEME_BENCH_E2E_DROPIN_EXE="$V" MEME_BENCH_E2E_STDERR=$O/e2e MEME_BENCH_PARITY_READS=50000 \
timeout 1300 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err
echo "bench rc $?" >> $O/bench.err
grep -E "e2e:|bench rc" $O/bench.err | cut -c1-150
rm -f $O/e2e/*.stderr      # (twelve copies of the per-chunk report: not needed, and they count against what is merged back)
ls $O/e2e | head -20; du -sh gpurun_out | tail -1
