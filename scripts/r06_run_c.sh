#!/bin/bash
# Round 6, GPU call C.  SURVEY 8(a) S1-S9 (SA-search stage, VERDICT item 6: experiments i and ii) + 8(b) per-read API row + the read-counter race test.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_perread.py tests/test_gpu_sam.py -x -q -m gpu -s > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rc |stale read counter|Error|error" $O/pytest.log | tail -12
timeout 1500 python scripts/r06_seed_variants.py 3100 10 "head=libmeme_hip.so,align=libmeme_hip_align.so,e4=libmeme_hip_e4.so" "28,24,22" $O/seed_variants.json > $O/seed_variants.log 2>&1; echo "variants rc $?" >> $O/seed_variants.log
grep -E "variants\]|rc " $O/seed_variants.log | cut -c1-420
du -sh gpurun_out | tail -1
