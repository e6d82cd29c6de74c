#!/bin/bash
# Round 6, GPU call N15 (the last GPU minute).  The determinism tests (36 varied arrangements + the read-counter regression test) on the very last binding.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06n15; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_determinism.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
