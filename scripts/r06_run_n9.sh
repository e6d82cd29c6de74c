#!/bin/bash
# Round 6, GPU call N9 (short).  bench.py's counter passes: the traffic with the passes taking the parent's reads (saved when they were sampled) against passes that
# sample again -- calls R2 / R3 reported twice the traffic of every earlier run with reads saved at the END of the run (a leg changes the host batch in place? the run now says so).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n9; mkdir -p $O
for cache in 1 0; do
MEME_BENCH_PMC_READS_CACHE=$cache MEME_BENCH_MBP=512 MEME_BENCH_READS=4000000 MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=1 MEME_BENCH_EXT=1 MEME_BENCH_C4=0 MEME_BENCH_RD=0 \
MEME_BENCH_PMC=1 MEME_BENCH_PARITY_READS=100000 timeout 400 python bench.py --steps 2 --warmup 1 > $O/bench_cache$cache.json 2> $O/bench_cache$cache.err; echo "rc $?" >> $O/bench_cache$cache.err
grep -E "no longer what was sampled|rc |pmc" $O/bench_cache$cache.err | cut -c1-200
python - <<PY
import json
d = json.loads(open("$O/bench_cache$cache.json").read().strip().split("\n")[-1])
r = d["roofline"]
print("cache=$cache traffic %.2f GB = %.2f x algorithmic" % (r["traffic"] / 1e9, r["traffic_over_algorithmic"]), r.get("traffic_counters_kb_per_launch"), "ext ok", d["ext"]["matches_oracle"], "chain ok", d["chain"]["matches_oracle"])
PY
done
