#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_bsw; rm -rf $OUT; mkdir -p $OUT
python scripts/bsw_probe.py 2 150 1 2>&1 | grep "bsw probe"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bsw -- python scripts/bsw_probe.py 2 150 1 > /dev/null 2> $OUT/p1.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -o bsw -- python scripts/bsw_probe.py 2 150 1 > /dev/null 2> $OUT/p2.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq2 -o bsw -- python scripts/bsw_probe.py 2 150 1 > /dev/null 2> $OUT/p3.err
for d in trace pmc_sq pmc_sq2; do python scripts/rocpd_summary.py $OUT/$d/bsw_results.db > $OUT/$d.md 2>&1; rm -rf $OUT/$d; done
cat $OUT/trace.md | head -20; grep -h "k_bsw_lane" $OUT/pmc_sq.md $OUT/pmc_sq2.md
