#!/bin/bash
# Round 6, GPU call O2 (the same soak once more, after the last changes of the round: CIGAR kernel row trim, bench trims).  The determinism soak on the round's last code (>= 200 varied runs identical): 128 Mbp genome, 300 k pairs,
# 216 runs over the 78 arrangements of scripts/r06_soak.py, each SAM against the unmodified reference's for its -K; then four runs under ThreadSanitizer.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06o2; mkdir -p $O
SOAK_TSAN=2 timeout 830 python scripts/r06_soak.py 128 0.3 216 $O/soak.json > $O/soak.log 2>&1; echo "soak rc $?" >> $O/soak.log
tail -8 $O/soak.log | cut -c1-300
cat $O/r06_tsan_*.* 2>/dev/null | grep -A12 "WARNING: ThreadSanitizer" | grep -E "WARNING|#0|#1" | sort | uniq -c | sort -rn | head -20 > $O/tsan_summary.txt
head -20 $O/tsan_summary.txt | cut -c1-220
