#!/bin/bash
# Round 6, GPU call B.  SURVEY 8 (b) / bar (1): ThreadSanitizer runs of the bound aligner (non-PIE + setarch -R after call A's "unexpected memory mapping"),
# and the drop-in SAM tests with the SAM-text stage refused whole / in pieces (advisor r05, medium).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06b; mkdir -p $O
SOAK_TSAN=4 timeout 3000 python scripts/r06_soak.py 128 0.3 4 $O/r06_soak.json > $O/soak.log 2>&1; echo "soak rc $?" >> $O/soak.log
grep -E "^index|^reference|^SOAK|^tsan|rc " $O/soak.log | cut -c1-600
for f in $O/r06_tsan_*; do [ -f "$f" ] && { echo "== $f"; grep -v "^\[" "$f" | head -c 5000; }; done 2>/dev/null | head -220
timeout 1500 python -m pytest tests/test_gpu_sam_e2e.py -x -q -m gpu -k "identical_to_reference" > $O/pytest_e2e.log 2>&1; echo "pytest rc $?" >> $O/pytest_e2e.log
tail -5 $O/pytest_e2e.log
du -sh gpurun_out | tail -1
