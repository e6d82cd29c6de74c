#!/bin/bash
MBP=${1:-512}
for v in "" _mw4 _e3 _mw4e3; do
  lib=$PWD/bwa-meme_amd/libmeme_hip$v.so
  [ -f $lib ] || continue
  echo "== variant ${v:-default}"
  bpc=5; case "$v" in _mw4*) bpc=4;; esac
  MEME_HIP_LIB=$lib LANES=${LANES:-4} BPC=$bpc timeout 600 python scripts/occ_probe.py $MBP 2 2>&1 | grep G=
done
