#!/bin/bash
# One gpurun call: bench at the named config (builds + caches the index in /dev/shm), then rocprofv3 passes that
# reuse the cache.  Outputs under gpurun_out/prof_named/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_named; mkdir -p $OUT
export MEME_BENCH_CPU=0
export MEME_BENCH_BSW=1
python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
MEME_BENCH_BSW=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o seed -- python bench.py --steps 3 --warmup 1 > /dev/null 2> $OUT/p1.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o seed -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $OUT/p2.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o seed -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $OUT/p3.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -o seed -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $OUT/p4.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $OUT/pmc_sq2 -o seed -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $OUT/p5.err
cat $OUT/bench.json
for d in trace pmc_fetch pmc_write pmc_sq pmc_sq2; do python scripts/rocpd_summary.py $OUT/$d/seed_results.db > $OUT/$d.md 2>&1; rm -rf $OUT/$d; done
grep -h "k_seed\|k_bsw" $OUT/*.md | head -60
echo "== reference CPU baseline at the named config"
MEME_BENCH_CPU=reference MEME_BENCH_CPU_READS=2000000 python bench.py --steps 2 --warmup 1 > $OUT/bench_refcpu.json 2> $OUT/bench_refcpu.err
grep cpu_baseline $OUT/bench_refcpu.err | tail -3
python -c "import json; d=json.load(open('$OUT/bench_refcpu.json')); print(d['value'], d['cpu_baseline'])"
