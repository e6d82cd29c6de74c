#!/bin/bash
# Round 6, GPU call I.  SURVEY 8(b)/(f)2: the tests call H failed (the mate-rescue verify shared a ctx with the CIGAR pre-pass's), three times over.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06i; mkdir -p $O
for k in 1 2 3; do
timeout 1500 python -m pytest tests/test_gpu_sam_e2e.py::test_sam_identical_to_reference tests/test_gpu_determinism.py tests/test_gpu_repeat_dense.py -q -m gpu -s > $O/pytest_$k.log 2>&1; echo "pytest rc $?" >> $O/pytest_$k.log
grep -E "passed|failed|rc |repeat-dense:|stale read|Memory access|MATE_CHECK" $O/pytest_$k.log | tail -6 | cut -c1-300
done
