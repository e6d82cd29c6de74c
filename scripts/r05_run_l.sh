#!/bin/bash
# Round 5, GPU call L.  SURVEY 8 (f)1 / B8: the extension stage in rounds after its overhead was cut (count arrays instead of an atomic per read,
# plan fused into k_ext_advance, four reads per workgroup) -- parity, both read classes, 1 / 2 / 3 / 4 one-seed rounds.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ext.py tests/test_gpu_sam.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
export MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=1 MEME_BENCH_C4=1 MEME_BENCH_C4_E2E=0 MEME_BENCH_PMC=0 MEME_BENCH_PARITY_READS=50000 MEME_BENCH_EXT_ROUNDS=1,2,3,4
export ROCPD_KERNELS=k_ext,k_bsw,k_chain,k_scan,k_flt ROCPD_ROWS=60
rocprofv3 --kernel-trace --stats -d $O/trace_ext -o ext -- python bench.py --steps 2 --warmup 1 > $O/bench_traced_ext.json 2> $O/p1.err
python scripts/rocpd_summary.py $O/trace_ext/ext_results.db > $O/trace_ext.md 2>&1; rm -rf $O/trace_ext
grep -h "k_ext\|k_bsw_lane\|k_scan" $O/trace_ext.md | head -20
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05l/bench_traced_ext.json').read().strip().splitlines()[-1])
print(json.dumps({x:d['ext'][x] for x in ('value','all_seeds_at_once','in_rounds') if x in d['ext']}))
c=d.get('config4_class',{})
if 'ext' in c: print('c4', json.dumps({x:c['ext'][x] for x in ('value','all_seeds_at_once','in_rounds') if x in c['ext']}))
PY
