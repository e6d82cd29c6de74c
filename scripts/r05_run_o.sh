#!/bin/bash
# Round 5, GPU call O.  SURVEY 8 (a) S13 (mem_chain_Learned on the device): what decides the stage's 7.4 ms -- the routed tiers' streams with and
# without the highest stream priority, the routing threshold swept; timeline of one call.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o; mkdir -p $O
export CHAIN_LIGHT_SWEEP=16,24,32,48,64,96,128
CHAIN_SIDE_PRIORITY=0 python scripts/chain_probe.py 3100 2 2>&1 | grep "chain probe" > $O/plain_p0.log
CHAIN_SIDE_PRIORITY=1 python scripts/chain_probe.py 3100 2 2>&1 | grep "chain probe" > $O/plain_p1.log
unset CHAIN_LIGHT_SWEEP
export ROCPD_ROWS=60 ROCPD_KERNELS=k_chain
CHAIN_SIDE_PRIORITY=1 rocprofv3 --kernel-trace --stats -d $O/t_chain -o c -- python scripts/chain_probe.py 3100 2 > /dev/null 2> $O/p1.err
python scripts/rocpd_timeline.py $O/t_chain/c_results.db k_chain 11 > $O/chain_timeline_p1.txt; rm -rf $O/t_chain
grep -h "chain kernels\|skip the lane" $O/plain_p0.log; echo ---; grep -h "chain kernels\|skip the lane" $O/plain_p1.log; cat $O/chain_timeline_p1.txt | cut -c1-150
