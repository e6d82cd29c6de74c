cd $GRAFT_REPO_ROOT
for i in 1 2; do
for m in "MEME_DROPIN_MATESW=0" "MEME_DROPIN_MATESW_PAR=0"; do
env $m E2E_SKIP_REF=1 E2E_READ_LEN=250 E2E_SUB=0.05 python scripts/e2e_bench.py 256 1 64 2>&1 | grep -E "WORKER_SAM|MEM_PROCESS_SEQ|whole pre-pass" | sed -e 's/.*WORKER_SAM avg: \([0-9.]*\).*/SAM \1/' -e 's/.*MEM_PROCESS_SEQ.*avg: \([0-9.]*\).*/PROC \1/' -e 's/.*mate rescue.*: \([0-9]*\) Smith.*kernels \([0-9.]*\) s, whole pre-pass \([0-9.]*\).*/MATE jobs \1 kern \2 pre \3/' -e 's/.*CIGAR.*whole pre-pass \([0-9.]*\).*/CIGPRE \1/' | tr '\n' ' '; echo " <- $m"
done; done
