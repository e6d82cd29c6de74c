#!/bin/bash
# Round 6, GPU call Q.  The whole -m gpu suite on the round's last code, as the driver runs it (plus durations), then smoke().
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06q; mkdir -p $O
timeout 1000 python -m pytest tests -q -m gpu --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc |^E  |Error" $O/pytest_gpu.log | tail -8 | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
