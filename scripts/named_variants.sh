#!/bin/bash
# named-configuration A/B in one gpurun call (index cached in /dev/shm between the runs)
cd $GRAFT_REPO_ROOT
export MEME_BENCH_CPU=0 MEME_BENCH_BSW=0
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e6,2), 'M reads/s; k_seed', round(d['roofline']['kernel_ms'],1), 'ms; windows/search', round(d['config']['windows_per_search'],3), 'leaves 2^%d' % d['config']['rmi_leaves_log2'])"; }
timeout 900 python bench.py --steps 3 --warmup 1 2>/dev/null | pr default
for lib in bwa-meme_amd/libmeme_hip_*.so; do
  [ -f $lib ] || continue
  MEME_HIP_LIB=$PWD/$lib timeout 600 python bench.py --steps 3 --warmup 1 2>/dev/null | pr $(basename $lib)
done
[ -n "$WITH_BITS30" ] && MEME_BENCH_BITS=30 timeout 900 python bench.py --steps 3 --warmup 1 2>/dev/null | pr bits30
