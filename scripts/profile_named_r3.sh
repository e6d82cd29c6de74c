#!/bin/bash
# Round-3 evidence at the named configuration, one gpurun call: a kernel trace of the bench command with its device legs (seeding,
# chain, ext + cigar, bsw; the CPU and e2e legs off), then separate --pmc passes (seeding only).  Outputs under gpurun_out/prof_named_r3/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_named_r3; mkdir -p $OUT
export MEME_BENCH_CPU=0 MEME_BENCH_E2E=0
rocprofv3 --kernel-trace --stats -d $OUT/trace -o seed -- python bench.py --steps 5 --warmup 1 > $OUT/bench_traced.json 2> $OUT/p1.err
export MEME_BENCH_BSW=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0
# seeding only: every k_seed dispatch but the 20 k-read parity sample is a 10 M-read launch (the average the bench line has to agree with)
rocprofv3 --kernel-trace --stats -d $OUT/trace_seed -o seed -- python bench.py --steps 5 --warmup 1 > $OUT/bench_traced_seed.json 2> $OUT/p1b.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o seed -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $OUT/p2.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o seed -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $OUT/p3.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -o seed -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $OUT/p4.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $OUT/pmc_sq2 -o seed -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $OUT/p5.err
for d in trace trace_seed pmc_fetch pmc_write pmc_sq pmc_sq2; do ROCPD_ROWS=40 python scripts/rocpd_summary.py $OUT/$d/seed_results.db > $OUT/$d.md 2>&1; rm -rf $OUT/$d; done
cut -c1-400 $OUT/bench_traced.json
grep -h "k_seed\|k_bsw_lane\|k_gather\|k_chain\|k_ext\|k_gcig" $OUT/*.md | head -80
