#!/bin/bash
# Round 5, GPU call H.  SURVEY 8 rows S7-S9 (third round of k_seed from plcp windows: parity with the switch on, then the named configuration's stage time on / off on one box).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05h; mkdir -p $O
MEME_TUNING=seed_r3_table=1 timeout 1200 python -m pytest tests/test_gpu_seed.py tests/test_gpu_scale.py -x -q -m gpu > $O/pytest_on.log 2>&1
echo "pytest rc $?" >> $O/pytest_on.log; tail -4 $O/pytest_on.log
export MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_PMC=0 MEME_BENCH_E2E=0 MEME_BENCH_PARITY_READS=1000000
for v in 1 0 1 0; do
  MEME_TUNING=seed_r3_table=$v timeout 900 python bench.py --steps 5 --warmup 1 > $O/bench_r3_$v.json 2> $O/bench_r3_$v.err
  python - <<PY
import json
d=json.loads(open('$O/bench_r3_$v.json').read().strip().split('\n')[-1])
print('r3_table=$v value',d['value'],'ms/step',d['ms_per_step'],'frac',d['roofline']['frac'],'stage',d['roofline']['kernel_ms'],'reseed',d['roofline']['of_which_reseed_kernels_ms'],'wps',d['config']['windows_per_search'],'parity',d['config']['sample_parity_with_oracle'])
PY
done
