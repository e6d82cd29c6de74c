cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_named_r3; mkdir -p $OUT
export MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0
rocprofv3 --kernel-trace --stats -d $OUT/trace_seed -o seed -- python bench.py --steps 5 --warmup 1 > $OUT/bench_traced_seed.json 2> $OUT/p1b.err
python scripts/rocpd_summary.py $OUT/trace_seed/seed_results.db > $OUT/trace_seed.md 2>&1; rm -rf $OUT/trace_seed
grep k_seed $OUT/trace_seed.md
