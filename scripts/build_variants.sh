#!/bin/bash
# build libmeme_hip_<name>.so variants: build_variants.sh name1="-DX=1 -DY=2" name2="..."
cd "$(dirname "$0")/../bwa-meme_amd"
for kv in "$@"; do
  name=${kv%%=*}; flags=${kv#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -shared -I../include -Icsrc csrc/*.hip -o libmeme_hip_$name.so &
done
wait
ls -la libmeme_hip_*.so
