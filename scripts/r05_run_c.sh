#!/bin/bash
# Round 5, GPU call C.  SURVEY 8: SAM identity (the first gate) -- run-to-run stability of the bound aligner after a differing md5 in call B.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c; mkdir -p $O
timeout 1700 python scripts/nondet_probe.py 128 1.5 5 > $O/nondet.log 2>&1
echo "rc $?" >> $O/nondet.log
tail -60 $O/nondet.log
