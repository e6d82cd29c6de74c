#!/bin/bash
# Round 5, GPU call V.  (1) SURVEY 8 (a) S13: is the chaining stage of the last code 7.4 ms (every run but one) or 9.1 ms (the final bench run)?  chain probe,
# three calls.  (2) SURVEY 8 (f)2: does k_gcig's 250-bp time follow the backtrack matrix's way to HBM?  The same 400 k calls with the matrix kept in LDS
# (gcig_zcap=28000: ~5 wavefronts per CU) against the default (2 KB window, matrix in HBM, ~28 per CU) and no LDS at all.  Probe mode: the ext leg reports no value.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05v; mkdir -p $O
timeout 400 python scripts/chain_probe.py 3100 2 2>&1 | grep "chain probe\] chain kernels\|chaining call" > $O/chain.log; cat $O/chain.log | cut -c1-200
export MEME_BENCH_MBP=256 MEME_BENCH_READS=2000000 MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=1 MEME_BENCH_C4_E2E=0 MEME_BENCH_PMC=0 MEME_BENCH_PARITY_READS=20000 MEME_BENCH_EXT_CHECK=0
for z in default 28000 0; do
  if [ $z = default ]; then T=""; else T="gcig_zcap=$z"; fi
  MEME_TUNING="$T" timeout 240 python bench.py --steps 2 --warmup 1 > $O/bench_z$z.json 2> $O/bench_z$z.err
  Z=$z python - <<'PY'
import json,os
z=os.environ['Z']
d=json.loads(open('gpurun_out/r05v/bench_z%s.json'%z).read().strip().splitlines()[-1])
c=d['config4_class']['ext']['cigar']; print('gcig_zcap',z,'| 250bp cigar ms',round(c['kernel_ms'],2),'| sampled check',c['checked'],'ok',c['matches_oracle'])
PY
done
