#!/bin/bash
# Round 6, GPU call N16 (the last seconds of the budget).  `mem -I`: insert sizes given, mem_pestat not called -- SAM against the reference's.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06n16; mkdir -p $O
timeout 40 python -m pytest tests/test_gpu_sam_e2e.py -x -q -m gpu -k "given_insert_size" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log | cut -c1-300
