#!/bin/bash
# Round 5, GPU call K.  SURVEY 8 (f)1 / B8: kernel trace of the extension stage (all seeds at once vs in rounds), 150-bp class -- where the
# stage's time outside banded SW goes.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ext.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
export MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=1 MEME_BENCH_C4=0 MEME_BENCH_PMC=0 MEME_BENCH_PARITY_READS=50000
export ROCPD_KERNELS=k_ext,k_bsw,k_chain,k_scan,k_flt ROCPD_ROWS=60
rocprofv3 --kernel-trace --stats -d $O/trace_ext -o ext -- python bench.py --steps 2 --warmup 1 > $O/bench_traced_ext.json 2> $O/p1.err
python scripts/rocpd_summary.py $O/trace_ext/ext_results.db > $O/trace_ext.md 2>&1; rm -rf $O/trace_ext
grep -h "k_ext\|k_bsw\|k_scan" $O/trace_ext.md | head -40
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05k/bench_traced_ext.json').read().strip().splitlines()[-1])
print(json.dumps({x:d['ext'][x] for x in ('value','all_seeds_at_once','in_rounds') if x in d['ext']}))
PY
