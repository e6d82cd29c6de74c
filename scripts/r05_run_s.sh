#!/bin/bash
# Round 5, GPU call S.  SURVEY 8 (f)2 (bwa_gen_cigar2 on the device): what a CIGAR job's time in k_gcig is made of -- resident wavefronts, instructions
# per job, issue vs wait -- one read class per counter pass, on a 256 Mbp probe genome with the ext leg's oracle pass off (MEME_BENCH_EXT_CHECK=0: that
# leg reports no value; the CIGAR sub-leg keeps its sampled check).  Raw profiler output stays in /tmp on the box: only summaries come back.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${S_OUT:-r05s}; mkdir -p $O; RAW=/tmp/r05s_raw; mkdir -p $RAW
timeout 600 python -m pytest ${S_TESTS:-tests/test_gpu_gcig.py} -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -2 $O/pytest.log
export MEME_BENCH_MBP=256 MEME_BENCH_READS=2000000 MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_C4_E2E=0 MEME_BENCH_PMC=0 MEME_BENCH_PARITY_READS=20000 MEME_BENCH_EXT_CHECK=0
export ROCPD_KERNELS=k_gcig ROCPD_ROWS=20
MEME_BENCH_EXT=1 MEME_BENCH_C4=1 timeout 240 python bench.py --steps 2 --warmup 1 > $O/bench_plain.json 2> $O/bench_plain.err
S_PLAIN=$O/bench_plain.json python - <<'PY'
import json
d=json.loads(open(__import__('os').environ.get('S_PLAIN')).read().strip().splitlines()[-1])
for name,e in (('150bp',d['ext']),('250bp',d['config4_class']['ext'])):
    c=e['cigar']; print('plain', name, 'cigar ms', round(c['kernel_ms'],2), 'checked', c['checked'], 'ok', c['matches_oracle'], '| ext leg value', e['value'], 'matches_oracle', e['matches_oracle'])
PY
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
for cls in ${S_CLASSES:-150 250}; do
  if [ $cls = 150 ]; then E=1; C=0; else E=0; C=1; fi
  for p in ${S_PASSES:-1 2}; do
    if [ $p = 1 ]; then CTR="$P1"; else CTR="$P2"; fi
    rm -rf $RAW/x
    MEME_BENCH_EXT=$E MEME_BENCH_C4=$C timeout 240 rocprofv3 --pmc $CTR -d $RAW/x -o t -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $O/pmc_${cls}_$p.err
    python scripts/rocpd_summary.py $RAW/x/t_results.db > $O/pmc_${cls}_$p.md 2>&1
    rm -rf $RAW/x
    echo "== ${cls}-bp class, pass $p"; grep -h "k_gcig(" $O/pmc_${cls}_$p.md | cut -c1-170
  done
done
du -sh gpurun_out | tail -1
