#!/bin/bash
# Round 6, GPU call G.  SURVEY 8(f)2: mate rescue posed on the device -- parity (fixture, oracle, in-aligner cross-check against the reference's own posing function in every paired SAM test) and a same-box
# A/B of the bound aligner at the named configuration (device posing vs the host's); the repeat-dense tests with the denser recipe.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06g; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_mate.py tests/test_gpu_repeat_dense.py tests/test_gpu_sam_e2e.py tests/test_gpu_determinism.py tests/test_gpu_kswv.py -x -q -m gpu -s > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rc |repeat-dense:|stale read|Error|assert|MATE_CHECK" $O/pytest.log | tail -12 | cut -c1-300
V="bwa-meme_dropin,bwa-meme_dropin@MEME_DROPIN_MATE_POSE=0,bwa-meme_dropin@X=2,bwa-meme_dropin@MEME_DROPIN_MATE_POSE=0@X=2"
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_RD=0 MEME_BENCH_PMC=0 MEME_BENCH_E2E_SKIP_REF=1 MEME_BENCH_E2E_PAIRS=4000000 MEME_BENCH_E2E_SLICES=0 \
MEME_BENCH_E2E_DROPIN_EXE="$V" MEME_BENCH_PARITY_READS=50000 MEME_BENCH_E2E_STDERR=$O/e2e timeout 1500 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
grep -E "e2e:|bench rc|failed" $O/bench.err | cut -c1-200
grep -h "mate rescue on the device" $O/e2e/*.stderr | tail -4 | cut -c1-400
rm -f $O/e2e/*X=2*.stderr
du -sh gpurun_out | tail -1
