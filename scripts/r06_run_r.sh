#!/bin/bash
# Round 6, GPU call R.  SURVEY 8(d): the whole default line -- `python bench.py`, no switches -- on the round's last code.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${R_OUT:-r06r}; mkdir -p $O
T0=$(date +%s)
MEME_BENCH_E2E_STDERR=$O/e2e timeout 1080 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $? wall $(( $(date +%s) - T0 )) s" >> $O/bench.err
grep -E "^\[bench|bench rc|failed" $O/bench.err | cut -c1-220
cut -c1-1500 $O/bench.json
