#!/bin/bash
# Round 5, GPU call Y.  SURVEY 8 (b): the binaries of the from-scratch build (library byte-identical to the incremental one; drop-in executables relinked) as the
# driver will use them at round end: __graft_entry__.smoke(), the SAM-identity tests that run the bound aligner, the extension and CIGAR fixtures.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05y; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke() ok')" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log; tail -3 $O/smoke.log
timeout 900 python -m pytest tests/test_gpu_sam_e2e.py tests/test_gpu_ext.py tests/test_gpu_gcig.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
