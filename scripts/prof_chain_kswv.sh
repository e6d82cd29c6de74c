#!/bin/bash
# Round-3 evidence for the two kernels added behind seeding this round: the chaining tiers and the mate-rescue kernel.  One gpurun call:
# plain runs, kernel traces, one --pmc pass each.  Outputs under gpurun_out/prof_ck/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_ck; rm -rf $OUT; mkdir -p $OUT
export ROCPD_ROWS=60 ROCPD_KERNELS=k_chain,k_kswv
python scripts/chain_probe.py 3100 2 2>&1 | grep "chain probe" > $OUT/chain_plain.log
python scripts/kswv_probe.py 30000 200000 2>&1 | grep "kswv probe" > $OUT/kswv_plain.log
rocprofv3 --kernel-trace --stats -d $OUT/t_chain -o c -- python scripts/chain_probe.py 3100 2 > /dev/null 2> $OUT/p1.err
python scripts/rocpd_timeline.py $OUT/t_chain/c_results.db k_chain 11 > $OUT/chain_timeline.txt
rocprofv3 --kernel-trace --stats -d $OUT/t_kswv -o c -- python scripts/kswv_probe.py 200000 > /dev/null 2> $OUT/p2.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY -d $OUT/p_kswv -o c -- python scripts/kswv_probe.py 200000 > /dev/null 2> $OUT/p3.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY -d $OUT/p_chain -o c -- python scripts/chain_probe.py 3100 2 > /dev/null 2> $OUT/p4.err
for d in t_chain t_kswv p_kswv p_chain; do python scripts/rocpd_summary.py $OUT/$d/c_results.db > $OUT/$d.md 2>&1; rm -rf $OUT/$d; done
cat $OUT/chain_plain.log $OUT/kswv_plain.log; grep -h "k_kswv" $OUT/p_kswv.md | head
