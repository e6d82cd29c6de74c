#!/bin/bash
# Round-6 evidence at the named configuration, one gpurun call: a kernel trace of the bench command (seeding only: the CPU, e2e, chain, ext
# and bsw legs off; the in-run parity check cut to one 50 000-read slice so that all other dispatches are 10 M-read launches), then separate
# --pmc passes.  The SA-search stage is now several kernels (k_seed<4>, k_reseed, k_reseed_emit, k_reseed_search, k_reseed_resume<..>): the
# tables list them all, scripts/make_named_profile_md_r5.py sums them.  Outputs under gpurun_out/prof_named_r6/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_named_r6; mkdir -p $OUT
export MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_RD=0 MEME_BENCH_PMC=0 MEME_BENCH_PARITY_READS=50000
export ROCPD_KERNELS=k_seed,k_reseed,k_gather,k_pack_reads ROCPD_ROWS=40
# Each pass writes its raw database under /tmp on the box and is summarised and deleted before the next one starts: a call that is cut short then still
# brings back what it finished (raw databases left under gpurun_out/ exceed the 64 MiB that are merged back -- that lost the counter passes of call R).
RAW=/tmp/prof_named_r6_raw
pass() {   # name, output redirection target for the bench line, steps, then the rocprofv3 options
    local name=$1 line=$2 steps=$3; shift 3
    rm -rf $RAW; mkdir -p $RAW
    rocprofv3 "$@" -d $RAW -o seed -- python bench.py --steps $steps --warmup 1 > $line 2> $OUT/$name.err
    python scripts/rocpd_summary.py $RAW/seed_results.db > $OUT/$name.md 2>&1
    rm -rf $RAW
}
pass trace_seed $OUT/bench_traced_seed.json 5 --kernel-trace --stats
pass pmc_fetch /dev/null 2 --pmc FETCH_SIZE
pass pmc_write /dev/null 2 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass pmc_sq /dev/null 2 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass pmc_sq2 /dev/null 2 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
cut -c1-600 $OUT/bench_traced_seed.json
grep -h "k_seed\|k_reseed\|k_gather\|k_build_plcp" $OUT/trace_seed.md | head -20
