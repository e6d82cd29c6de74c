#!/bin/bash
# Round 5, GPU call F.  SURVEY 8 rows served: (f)4 (SAM text on the device: kernel parity, SAM identity with the in-aligner cross-check, A/B on one box),
# S7-S10 (the overflow tier beside the re-seeding batches: seeding parity + the named configuration's stage time).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05f; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sam.py tests/test_gpu_gcig.py tests/test_gpu_seed.py tests/test_gpu_scale.py tests/test_gpu_sam_e2e.py tests/test_gpu_sam_scale.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_E2E_PAIRS=2000000 MEME_BENCH_E2E_SKIP_REF=1 MEME_BENCH_E2E_KEEP_DIFF=1 \
MEME_BENCH_E2E_DROPIN_EXE="bwa-meme_dropin,bwa-meme_dropin@MEME_DROPIN_SAM=0,r04/bwa-meme_dropin_r04,bwa-meme_dropin@X=1,bwa-meme_dropin@MEME_DROPIN_SAM=0@X=1,bwa-meme_dropin_prof" \
MEME_BENCH_E2E_STDERR=$O/e2e MEME_BENCH_PARITY_READS=200000 MEME_SEED_TRACE=1 \
timeout 1500 python bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err
echo "bench rc $?" >> $O/bench.err
grep -E "e2e:|seed tier|parity" $O/bench.err | tail -24
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05f/bench.json').read().strip().split('\n')[-1])
print('value',d['value'],'ms/step',d['ms_per_step'],'frac',d['roofline']['frac'],'stage',d['roofline']['kernel_ms'],'reseed',d['roofline']['of_which_reseed_kernels_ms'])
PY
