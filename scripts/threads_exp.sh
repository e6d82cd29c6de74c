cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_sam_e2e.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
for m in "MEME_DROPIN_MATESW=0" "MEME_DROPIN_MATESW=1"; do
env $m E2E_SKIP_REF=1 E2E_CHUNK=default python scripts/e2e_bench.py 256 2 64 2>&1 | grep -E "rc 0|WORKER_SAM|MEM_PROCESS_SEQ|whole pre-pass|md5" | sed -e 's/.*WORKER_SAM avg: \([0-9.]*\).*/SAM \1/' -e 's/.*MEM_PROCESS_SEQ.*avg: \([0-9.]*\).*/PROC \1/' -e 's/.*mate rescue.*: \([0-9]*\) Smith.*kernels \([0-9.]*\) s, whole pre-pass \([0-9.]*\).*/MATE jobs \1 kern \2 pre \3/' -e 's/.*CIGAR.*whole pre-pass \([0-9.]*\).*/CIGPRE \1/' -e "s/.*device': '\([0-9a-f]*\)'.*/\1/" -e 's/.*rc 0 wall \([0-9.]*\) s.*/WALL \1/' | tr '\n' ' '; echo " <- $m"
done; done
