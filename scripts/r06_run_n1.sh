#!/bin/bash
# Round 6, GPU call N1 (short on purpose: calls M and M2 lost their boxes after 868 s each with nothing returned).  k_gcig_grp against the fixtures, the oracle and
# k_gcig; the lower-case SAM test (the binding's 64-letter conversion); the first SAM test (one walk over the records); the CIGAR leg's times.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06n1; mkdir -p $O
(free -g | head -2; nproc; uptime) > $O/box.txt 2>&1
timeout 200 python -m pytest tests/test_gpu_gcig.py -x -q -m gpu > $O/pytest_gcig.log 2>&1; echo "pytest rc $?" >> $O/pytest_gcig.log; tail -3 $O/pytest_gcig.log
timeout 280 python -m pytest tests/test_gpu_sam_e2e.py -x -q -m gpu -k "lower_case or identical_to_reference" > $O/pytest_sam.log 2>&1; echo "pytest rc $?" >> $O/pytest_sam.log; tail -3 $O/pytest_sam.log
(uptime; free -g | head -2) >> $O/box.txt 2>&1
