#!/bin/bash
# Start-up of the bound aligner at GRCh38 size, several launches on ONE box: bash scripts/startup_probe.sh [variant ...]
# (variants: blocking_off prefetch_off io_off early_off; none = just a first and a second launch at the default settings).
# COST: the first invocation runs the unmodified reference too (~5.5 min of box time); every further one ~2.5 min -- but only because
# MEME_BENCH_E2E_REUSE_REF=1 lets an N=1 bench run reuse the reference's cached timing.  Without it every invocation re-runs the
# reference (185 s each): the first version of this script assumed the cache was read by N=1 runs, was killed at its 30-minute limit
# after four of seven variants and cost 30 GPU-minutes (profiles/r04_startup_probe.md).
export MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_PARITY_READS=20000
mkdir -p gpurun_out/startup
run() {  # name, env...
  name=$1; shift
  mkdir -p gpurun_out/startup/$name
  t0=$(date +%s)
  env "$@" MEME_BENCH_E2E_STDERR=gpurun_out/startup/$name python bench.py --steps 3 --warmup 1 > gpurun_out/startup/$name/bench.json 2> gpurun_out/startup/$name/bench.stderr
  python scripts/startup_probe_summary.py "$name" "$(( $(date +%s) - t0 ))"
}
run first_launch A=1
export MEME_BENCH_E2E_REUSE_REF=1
run second_launch A=1
for v in "$@"; do
  case $v in
    blocking_off) run $v MEME_DROPIN_BLOCKING_SYNC=0 ;;
    prefetch_off) run $v MEME_DROPIN_PREFETCH=0 ;;
    io_off)       run $v MEME_DROPIN_IO=0 ;;
    early_off)    run $v MEME_DROPIN_EARLY=0 ;;
  esac
done
