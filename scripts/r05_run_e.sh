#!/bin/bash
# Round 5, GPU call E.  SURVEY 8 (d): the whole default bench.py line (BASELINE configs[1] headline + configs[2] at 10 M pairs + configs[4]-class leg) -- timing of
# the legs against the driver's limit; then the whole -m gpu suite.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05e; mkdir -p $O
( time MEME_BENCH_E2E_STDERR=$O/e2e timeout 2400 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
echo "bench rc $?" >> $O/bench.err
grep -E "e2e:|leg failed|skipped|parity" $O/bench.err | tail -20; cat $O/bench.time
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
