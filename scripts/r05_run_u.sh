#!/bin/bash
# Round 5, GPU call U (the round's last code; same steps as call M).  SURVEY 8 (d): the whole default bench.py line with the live PMC passes (BASELINE configs[1] headline + configs[2] at 10 M pairs +
# configs[4]-class leg), the whole -m gpu suite, then the named-configuration kernel trace + counter passes (scripts/profile_named_r5.sh).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05u; mkdir -p $O
( time MEME_BENCH_E2E_STDERR=$O/e2e timeout 2700 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
echo "bench rc $?" >> $O/bench.err
grep -E "e2e:|leg failed|skipped|parity|pmc|traffic" $O/bench.err | tail -20; cat $O/bench.time
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 1500 bash scripts/profile_named_r5.sh > $O/profile_named.log 2>&1
tail -12 $O/profile_named.log | cut -c1-400
