#!/bin/bash
# Round 6, GPU call N6 (under ten minutes).  SURVEY 8(b), the bound aligner on a host with a 16-CPU quota: worker threads (-t) against process CPU and mem_process_seqs --
# 4 M pairs, GRCh38-sized index, every run's SAM md5 against the first.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06n6; mkdir -p $O
V="bwa-meme_dropin,bwa-meme_dropin@MEME_T=16,bwa-meme_dropin@MEME_T=24,bwa-meme_dropin@MEME_T=32,bwa-meme_dropin@MEME_T=48,bwa-meme_dropin@MEME_T=64@X=2,bwa-meme_dropin@MEME_T=24@X=2,bwa-meme_dropin@MEME_T=16@X=2,bwa-meme_dropin@MEME_T=32@X=2"
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_RD=0 MEME_BENCH_PMC=0 MEME_BENCH_E2E_SKIP_REF=1 MEME_BENCH_E2E_PAIRS=4000000 MEME_BENCH_E2E_SLICES=0 \
MEME_BENCH_E2E_DROPIN_EXE="$V" MEME_BENCH_PARITY_READS=50000 timeout 500 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
grep -E "e2e:|bench rc|failed" $O/bench.err | cut -c1-200
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().split("\n")[-1])["e2e"]
print({k: d.get(k) for k in ("sam_md5_dropin", "extra_runs_sam_identical")})
for k, v in (d.get("extra_runs") or {}).items():
    print(k, {x: v.get(x) for x in ("wall_s", "process_s", "cpu_s", "sam_md5")})
PY
