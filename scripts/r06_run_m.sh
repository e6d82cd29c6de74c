#!/bin/bash
# Round 6, GPU call M.  The binding with the letters->codes conversion 64 bytes at a time and ONE walk over the chunk's alignment records (digest):
# the SAM tests (incl. the new lower-case / ambiguity-letter case), the determinism test, then call L2's end-to-end run for the helper team's CPU by host loop.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sam_e2e.py tests/test_gpu_determinism.py tests/test_gpu_dropin.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
V="bwa-meme_dropin"
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_RD=0 MEME_BENCH_PMC=0 MEME_BENCH_E2E_SKIP_REF=1 MEME_BENCH_E2E_PAIRS=4000000 MEME_BENCH_E2E_SLICES=0 \
MEME_BENCH_E2E_DROPIN_EXE="$V" MEME_BENCH_PARITY_READS=50000 MEME_BENCH_E2E_STDERR=$O/e2e timeout 1200 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
grep -E "e2e:|bench rc|failed" $O/bench.err | cut -c1-200
for f in $O/e2e/*.stderr; do echo "== $f"; grep -h "by thread role\|by host loop" $f | cut -c1-1400; done
