#!/bin/bash
# Round 6, GPU call A.  SURVEY 8 (b) boundary / bar (1) bit-identical SAM: the determinism question of VERDICT r05 (one differing md5 in 36 runs).
# (1) tests/test_gpu_determinism.py: 30 varied runs incl. MEME_DROPIN_VERIFY (every device stage twice on two ctxs, compared in the aligner);
# (2) scripts/r06_soak.py: 128 Mbp paired workload, varied arrangements, SAM vs the unmodified reference, + ThreadSanitizer runs of the bound aligner.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_determinism.py -x -q -m gpu > $O/pytest_det.log 2>&1; echo "pytest rc $?" >> $O/pytest_det.log
tail -5 $O/pytest_det.log
SOAK_TSAN=${SOAK_TSAN:-3} timeout 2400 python scripts/r06_soak.py 128 0.3 ${SOAK_RUNS:-45} $O/r06_soak.json > $O/soak.log 2>&1; echo "soak rc $?" >> $O/soak.log
grep -E "^index|^reference|^SOAK|^tsan|rc " $O/soak.log | cut -c1-400
grep -c differing_lines $O/soak.log
for f in $O/r06_tsan_*; do [ -f "$f" ] && { echo "== $f"; head -c 6000 "$f"; }; done 2>/dev/null | head -150
du -sh gpurun_out | tail -1
