#!/bin/bash
# counter passes over scripts/quick_probe.py (rounds 1-3 only).  usage: prof_quick.sh [Mbp] [Mreads] [bits] [tag]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
MBP=${1:-512}; MR=${2:-4}; BITS=${3:-0}; TAG=${4:-quick}
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
export PROBE_ROUNDS=3
python scripts/quick_probe.py $MBP $MR $BITS 4 2>&1 | grep "rounds=" | tee $OUT/probe.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -o seed -- python scripts/quick_probe.py $MBP $MR $BITS 4 > /dev/null 2> $OUT/p4.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $OUT/pmc_sq2 -o seed -- python scripts/quick_probe.py $MBP $MR $BITS 4 > /dev/null 2> $OUT/p5.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o seed -- python scripts/quick_probe.py $MBP $MR $BITS 4 > /dev/null 2> $OUT/p2.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o seed -- python scripts/quick_probe.py $MBP $MR $BITS 4 > /dev/null 2> $OUT/p3.err
for d in pmc_sq pmc_sq2 pmc_fetch pmc_write; do python scripts/rocpd_summary.py $OUT/$d/seed_results.db > $OUT/$d.md 2>&1; rm -rf $OUT/$d; done
grep -h "k_seed" $OUT/*.md
