#!/bin/bash
# Round 6, GPU call N4 (under thirteen minutes).  The rest of the -m gpu suite on the last code (N3 ran the CIGAR / SAM / determinism / mate / extension files), then
# instruction counters of the CIGAR kernels with 4 / 2 / 1 jobs per wavefront and with one wavefront per job (150-bp class, 400 k calls, 256-Mbp probe genome).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n4; mkdir -p $O; RAW=/tmp/r06n4_raw; mkdir -p $RAW
timeout 540 python -m pytest tests/test_gpu_bsw.py tests/test_gpu_chain.py tests/test_gpu_kswv.py tests/test_gpu_multidev.py tests/test_gpu_perread.py tests/test_gpu_prmi.py tests/test_gpu_repeat_dense.py tests/test_gpu_sa.py \
  tests/test_gpu_sam.py tests/test_gpu_seed.py tests/test_gpu_scale.py tests/test_gpu_sam_scale.py tests/test_gpu_bench_multirank.py -q -m gpu --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rc |^E  |Error" $O/pytest.log | tail -8 | cut -c1-300
export MEME_BENCH_MBP=256 MEME_BENCH_READS=2000000 MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_C4_E2E=0 MEME_BENCH_RD=0 MEME_BENCH_PMC=0 MEME_BENCH_PARITY_READS=20000 MEME_BENCH_EXT_CHECK=0
export ROCPD_KERNELS=k_gcig ROCPD_ROWS=20
CTR="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
for gr in 1 0; do
  rm -rf $RAW/x
  MEME_TUNING="gcig_groups=$gr" MEME_BENCH_EXT=1 MEME_BENCH_C4=0 timeout 150 rocprofv3 --pmc $CTR -d $RAW/x -o t -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $O/pmc_150_groups$gr.err
  python scripts/rocpd_summary.py $RAW/x/t_results.db > $O/pmc_150_groups$gr.md 2>&1
  rm -rf $RAW/x
  echo "== 150-bp class, gcig_groups=$gr"; grep -h "k_gcig" $O/pmc_150_groups$gr.md | cut -c1-220 | head -12
done
du -sh gpurun_out | tail -1
