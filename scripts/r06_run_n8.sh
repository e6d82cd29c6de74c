#!/bin/bash
# Round 6, GPU call N8 (short).  SURVEY 8(f)2: why the two-columns-per-lane CIGAR kernel is not faster than two 64-column chunks (51.0 vs 47.8 ms per 400 k calls of the
# 250-bp class although its row loop has half the instructions): instruction, LDS-conflict and wait counters of both, 250-bp class only, 256-Mbp probe genome.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n8; mkdir -p $O; RAW=/tmp/r06n8_raw; mkdir -p $RAW
export MEME_BENCH_MBP=256 MEME_BENCH_READS=2000000 MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_C4_E2E=0 MEME_BENCH_RD=0 MEME_BENCH_PMC=0 MEME_BENCH_PARITY_READS=20000 MEME_BENCH_EXT_CHECK=0
export ROCPD_KERNELS=k_gcig ROCPD_ROWS=60
CTR="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
for gr in ${N8_VARIANTS:-1 0}; do
  rm -rf $RAW/x
  MEME_TUNING="gcig_groups=$gr" MEME_BENCH_EXT=0 MEME_BENCH_C4=1 timeout 200 rocprofv3 --pmc $CTR -d $RAW/x -o t -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $O/pmc_250_groups$gr.err
  python scripts/rocpd_summary.py $RAW/x/t_results.db > $O/pmc_250_groups$gr.md 2>&1
  rm -rf $RAW/x
  echo "== 250-bp class, gcig_groups=$gr"; grep -h "k_gcig_t" $O/pmc_250_groups$gr.md | cut -c1-200 | head -24
done
