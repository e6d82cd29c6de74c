#!/bin/bash
# Round 5, GPU call B.  SURVEY 8 rows served: (f)2 (the gap-free shortcut as a lane-per-job kernel: parity tests again), (f)4 (A/B of the bound
# aligner on ONE box: round 4's build, HEAD, HEAD with the CIGAR stage off, HEAD with SAM-phase timers).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gcig.py tests/test_gpu_sam_e2e.py -x -q -m gpu -k "gcig or gen_cigar or mixed_250bp or smart_pairing or identical_to_reference" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_E2E_PAIRS=2000000 MEME_BENCH_E2E_SKIP_REF=1 \
MEME_BENCH_E2E_DROPIN_EXE="bwa-meme_dropin,r04/bwa-meme_dropin_r04,bwa-meme_dropin@MEME_DROPIN_CIGAR=0,bwa-meme_dropin_prof,r04/bwa-meme_dropin_r04@X=2,bwa-meme_dropin@X=2" \
MEME_BENCH_E2E_STDERR=$O/e2e MEME_BENCH_PARITY_READS=50000 \
timeout 1500 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err
echo "bench rc $?" >> $O/bench.err
tail -3 $O/pytest.log; grep "e2e:" $O/bench.err
