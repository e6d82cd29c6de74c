#!/bin/bash
# Round 5, GPU call D.  SURVEY 8: SAM identity at GRCh38 size -- which records differ between runs of the same binary (call B saw one md5 of six differ);
# (f)4: OpenMP wait policy of the binding's helper teams under the boxes' CPU quota.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05d; mkdir -p $O
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_E2E_PAIRS=2000000 MEME_BENCH_E2E_SKIP_REF=1 MEME_BENCH_E2E_KEEP_DIFF=1 \
MEME_BENCH_E2E_DROPIN_EXE="bwa-meme_dropin,bwa-meme_dropin@X=1,bwa-meme_dropin@X=2,bwa-meme_dropin@X=3,bwa-meme_dropin@OMP_WAIT_POLICY=passive,bwa-meme_dropin@X=5,bwa-meme_dropin@MEME_DROPIN_CIGAR=0,bwa-meme_dropin@X=7,bwa-meme_dropin@OMP_WAIT_POLICY=passive@X=8,r04/bwa-meme_dropin_r04,bwa-meme_dropin@X=10" \
MEME_BENCH_E2E_STDERR=$O/e2e MEME_BENCH_PARITY_READS=50000 \
timeout 1700 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err
echo "bench rc $?" >> $O/bench.err
grep "e2e:" $O/bench.err; ls $O/e2e | grep diff
