#!/bin/bash
# Round 6, GPU call N12 (the round's last GPU minutes).  bench.py at the named configuration with every leg off but the counter passes: do they now run at the named
# size beside their parent (call R6: the parent still held 146 GB -- the device-built index's tensors under other names -- and the passes stopped, as they now should)?
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06n12; mkdir -p $O
MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_RD=0 MEME_BENCH_PMC=1 MEME_BENCH_PARITY_READS=50000 \
timeout 300 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "rc $?" >> $O/bench.err
grep -E "counter passes|pmc|rc " $O/bench.err | cut -c1-200
python - <<PY
import json
q = json.loads(open("$O/bench.json").read().strip().split("\n")[-1]); r = q["roofline"]
print("traffic %.1f GB = %.2f x" % (r["traffic"] / 1e9, r.get("traffic_over_algorithmic") or 0), r.get("traffic_source", "")[:30], r.get("frac_of_random_line_ceiling"))
print({k: v for k, v in (r.get("traffic_counter_passes") or {}).items() if k.endswith("pass_work")})
PY
