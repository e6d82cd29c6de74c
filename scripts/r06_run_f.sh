#!/bin/bash
# Round 6, GPU call F.  The whole -m gpu suite on the code so far (SURVEY 8 rows a-f: every parity test), incl. the round's new tests.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06f; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -40 $O/pytest_gpu.log | cut -c1-250
