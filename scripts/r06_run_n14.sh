#!/bin/bash
# Round 6, GPU call N14 (one minute).  The record digest built under a lock: a paired SAM run in which mem_pestat's hook does not build it first (MEME_DROPIN_PESTAT=0),
# so that the CIGAR and mate-rescue pre-passes of a half reach chunk_digest() side by side -- the path a run with -I takes.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06n14; mkdir -p $O
MEME_DROPIN_PESTAT=0 timeout 110 python -m pytest tests/test_gpu_sam_e2e.py -x -q -m gpu -k "identical_to_reference and True" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
