#!/bin/bash
# Round 5, GPU call P.  SURVEY 8 (f)2 (bwa_gen_cigar2 on the device): k_gcig with the target bases and small backtrack matrices in LDS -- parity
# (gcig fixtures, SAM text, SAM identity), the CIGAR figures of both read classes, e2e of the 250-bp class where the device is level with the host.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05p; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_gcig.py tests/test_gpu_sam.py tests/test_gpu_sam_e2e.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_PMC=0 MEME_BENCH_E2E_PAIRS=4000000 MEME_BENCH_E2E_SKIP_REF=1 MEME_BENCH_E2E_KEEP_DIFF=1 \
MEME_BENCH_E2E_DROPIN_EXE="bwa-meme_dropin,bwa-meme_dropin@X=1,r04/bwa-meme_dropin_r04" \
MEME_BENCH_E2E_STDERR=$O/e2e MEME_BENCH_PARITY_READS=50000 \
timeout 2400 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err
echo "bench rc $?" >> $O/bench.err
grep -E "e2e:" $O/bench.err | tail -30
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05p/bench.json').read().strip().splitlines()[-1])
print('cigar', json.dumps(d['ext']['cigar']))
c=d.get('config4_class',{})
print('c4 cigar', json.dumps(c['ext']['cigar']))
for k in ('e2e',):
    e=d[k]; print(k, e['dropin']['process_s'], json.dumps(e['dropin']['backend']))
    e=c[k]; print('c4',k, e['dropin']['process_s'], json.dumps(e['dropin']['backend']))
PY
