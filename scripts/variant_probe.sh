#!/bin/bash
# compare kernel variants on a mid-size index (512 Mbp: 16 GB of entries, beyond the 256 MB Infinity Cache)
MBP=${1:-512}
for v in "" _dyn _cmp4 _mw6 _mw4; do
  lib=$PWD/bwa-meme_amd/libmeme_hip$v.so
  [ -f $lib ] || continue
  echo "== variant ${v:-default}"
  bpc=5; [ "$v" = "_mw6" ] && bpc=6; [ "$v" = "_mw4" ] && bpc=4
  MEME_HIP_LIB=$lib LANES=4,8 BPC=$bpc timeout 600 python scripts/occ_probe.py $MBP 2 2>&1 | grep G=
done
