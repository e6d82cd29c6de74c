#!/bin/bash
# Round 6, GPU call N13 (one minute).  The binding rebuilt after letters_to_codes moved into a header of its own (no change of code): the SAM tests that run through it.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06n13; mkdir -p $O
timeout 150 python -m pytest tests/test_gpu_sam_e2e.py -x -q -m gpu -k "lower_case or identical_to_reference" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
