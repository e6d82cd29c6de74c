#!/bin/bash
# calibrate FETCH_SIZE / TCC counters on random gathers of known size (scripts/microbench/gather_roofline)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_gather; rm -rf $OUT; mkdir -p $OUT
timeout 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc -o g -- scripts/microbench/gather_roofline 8 16 > $OUT/run.log 2>&1
python scripts/rocpd_summary.py $OUT/pmc/g_results.db > $OUT/pmc.md 2>&1; rm -rf $OUT/pmc
cat $OUT/run.log | grep gather; cat $OUT/pmc.md
