#!/bin/bash
# run quick_probe for the default library and every libmeme_hip_*.so variant: run_variants.sh [Mbp] [Mreads] [bits] [lanes]
MBP=${1:-512}; MR=${2:-4}; BITS=${3:-0}; LANES=${4:-4}
python scripts/quick_probe.py $MBP $MR $BITS $LANES 2>&1 | grep "G=\|rror"
for lib in $PWD/bwa-meme_amd/libmeme_hip_*.so; do
  [ -f $lib ] || continue
  MEME_HIP_LIB=$lib timeout 300 python scripts/quick_probe.py $MBP $MR $BITS $LANES 2>&1 | grep "G=\|rror"
done
