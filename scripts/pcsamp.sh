#!/bin/bash
# PC sampling of the seeding kernel (needs a library built with -gline-tables-only): pcsamp.sh <lib> [method] [Mbp] [Mreads]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
LIB=$1; METHOD=${2:-host_trap}; MBP=${3:-512}; MR=${4:-4}
OUT=gpurun_out/pcsamp; rm -rf $OUT; mkdir -p $OUT
python scripts/quick_probe.py $MBP $MR 0 4 2>&1 | grep "G="     # builds / caches the index
if [ "$METHOD" = stochastic ]; then ARGS="--pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576"
else ARGS="--pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 100"; fi
MEME_HIP_LIB=$LIB timeout 600 rocprofv3 --pc-sampling-beta-enabled $ARGS --output-format csv -d $OUT/raw -o seed -- python scripts/quick_probe.py $MBP $MR 0 4 > $OUT/run.log 2>&1
tail -5 $OUT/run.log
find $OUT/raw -type f | head -20
python scripts/pcsamp_summary.py $OUT/raw > $OUT/summary.txt 2>&1
rm -rf $OUT/raw
head -150 $OUT/summary.txt
