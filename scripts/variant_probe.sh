#!/bin/bash
MBP=${1:-512}
for v in "" _ee2; do
  lib=$PWD/bwa-meme_amd/libmeme_hip$v.so
  [ -f $lib ] || continue
  echo "== variant ${v:-default}"
  MEME_HIP_LIB=$lib LANES=${LANES:-4,8} BPC=5 timeout 600 python scripts/occ_probe.py $MBP 2 2>&1 | grep G=
done
