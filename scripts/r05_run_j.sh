#!/bin/bash
# Round 5, GPU call J.  SURVEY 8 (f)1 / B8 (mem_chain2aln_across_reads_V2 on the device): extension in rounds + surviving records only --
# parity (ext fixtures, SAM identity), the ext legs of both read classes, e2e A/B on one box.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ext.py tests/test_gpu_sam_e2e.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_PMC=0 MEME_BENCH_E2E_PAIRS=4000000 MEME_BENCH_E2E_SKIP_REF=1 MEME_BENCH_E2E_KEEP_DIFF=1 \
MEME_BENCH_E2E_DROPIN_EXE="bwa-meme_dropin,bwa-meme_dropin@MEME_DROPIN_EXT_LIVE=0,bwa-meme_dropin@X=1,bwa-meme_dropin@MEME_DROPIN_EXT_LIVE=0@X=1,r04/bwa-meme_dropin_r04" \
MEME_BENCH_E2E_STDERR=$O/e2e MEME_BENCH_PARITY_READS=50000 \
timeout 2400 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err
echo "bench rc $?" >> $O/bench.err
grep -E "e2e:|ext|rounds" $O/bench.err | tail -30
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05j/bench.json').read().strip().splitlines()[-1])
for k in ('ext',):
    print(k, json.dumps({x:d[k][x] for x in ('value','all_seeds_at_once','in_rounds') if x in d[k]}))
c=d.get('config4_class',{})
if 'ext' in c: print('c4 ext', json.dumps({x:c['ext'][x] for x in ('value','all_seeds_at_once','in_rounds') if x in c['ext']}))
if 'e2e' in c: print('c4 e2e', json.dumps(c['e2e'])[:600])
PY
