#!/bin/bash
# Round 6, GPU call M (second attempt: the first lost its box after 12 minutes with nothing returned).  The binding with the letters->codes conversion 64 bytes
# at a time and ONE walk over the chunk's alignment records (digest); k_gcig_grp (several CIGAR jobs per wavefront).  Short steps, each under its own timeout.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06m; mkdir -p $O
free -g | head -2 > $O/mem.txt; nproc >> $O/mem.txt
timeout 600 python -m pytest tests/test_gpu_gcig.py -x -q -m gpu > $O/pytest_gcig.log 2>&1; echo "pytest rc $?" >> $O/pytest_gcig.log; tail -3 $O/pytest_gcig.log
timeout 1500 python -m pytest tests/test_gpu_sam_e2e.py -x -q -m gpu > $O/pytest_sam.log 2>&1; echo "pytest rc $?" >> $O/pytest_sam.log; tail -3 $O/pytest_sam.log
timeout 900 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_ext.py -x -q -m gpu > $O/pytest_det.log 2>&1; echo "pytest rc $?" >> $O/pytest_det.log; tail -3 $O/pytest_det.log
free -g | head -2 >> $O/mem.txt
