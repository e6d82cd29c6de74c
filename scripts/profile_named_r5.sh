#!/bin/bash
# Round-5 evidence at the named configuration, one gpurun call: a kernel trace of the bench command (seeding only: the CPU, e2e, chain, ext
# and bsw legs off; the in-run parity check cut to one 50 000-read slice so that all other dispatches are 10 M-read launches), then separate
# --pmc passes.  The SA-search stage is now several kernels (k_seed<4>, k_reseed, k_reseed_emit, k_reseed_search, k_reseed_resume<..>): the
# tables list them all, scripts/make_named_profile_md_r5.py sums them.  Outputs under gpurun_out/prof_named_r5/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_named_r5; mkdir -p $OUT
export MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_PMC=0 MEME_BENCH_PARITY_READS=50000
export ROCPD_KERNELS=k_seed,k_reseed,k_gather,k_pack_reads ROCPD_ROWS=40
rocprofv3 --kernel-trace --stats -d $OUT/trace_seed -o seed -- python bench.py --steps 5 --warmup 1 > $OUT/bench_traced_seed.json 2> $OUT/p1.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o seed -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $OUT/p2.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o seed -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $OUT/p3.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -o seed -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $OUT/p4.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $OUT/pmc_sq2 -o seed -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $OUT/p5.err
for d in trace_seed pmc_fetch pmc_write pmc_sq pmc_sq2; do python scripts/rocpd_summary.py $OUT/$d/seed_results.db > $OUT/$d.md 2>&1; rm -rf $OUT/$d; done
cut -c1-600 $OUT/bench_traced_seed.json
grep -h "k_seed\|k_reseed\|k_gather\|k_build_plcp" $OUT/trace_seed.md | head -20
