#!/bin/bash
# Round 6, GPU call N11 (short).  Why bench.py's counter passes do 5 % less work per read than the run that starts them (call R5: 8.17 vs 8.63 SMEMs per read): the same
# seeding-only command (one step) from the repository and from /tmp, alone and beside a process that holds 130 GB of HBM, each printing its own work counters.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06n11; mkdir -p $O
export MEME_BENCH_PMC=0 MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_RD=0 MEME_BENCH_PARITY_READS=0
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1]); c = d["config"]
    print(sys.argv[1], {k: c.get(k) for k in ("smems_per_read", "hits_per_read", "searches_per_read")}, "stage ms %.2f" % d["roofline"]["kernel_ms"])
except Exception as e:
    print(sys.argv[1], "no line", e)
PY
}
timeout 200 python bench.py --steps 1 --warmup 0 > $O/a_repo.json 2> $O/a_repo.err; show "from the repository:" $O/a_repo.json
(cd /tmp && timeout 200 python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 > $GRAFT_REPO_ROOT/$O/b_tmp.json 2> $GRAFT_REPO_ROOT/$O/b_tmp.err); show "from /tmp:" $O/b_tmp.json
python -c "
import torch, time
x = torch.empty(130 * (1 << 30), dtype=torch.uint8, device='cuda:0'); torch.cuda.synchronize(); print('holding 130 GB', flush=True); time.sleep(170)
" > $O/hog.log 2>&1 &
HOG=$!
sleep 25
timeout 200 python bench.py --steps 1 --warmup 0 > $O/c_hog.json 2> $O/c_hog.err; show "beside a process holding 130 GB:" $O/c_hog.json
kill $HOG 2>/dev/null; wait $HOG 2>/dev/null
tail -2 $O/c_hog.err | cut -c1-200
