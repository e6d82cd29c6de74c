#!/bin/bash
# Round 6, GPU call P.  SURVEY 8(d): kernel trace + counter passes at the named configuration on the round's last code (scripts/profile_named_r6.sh).
cd "$GRAFT_REPO_ROOT"
timeout 760 bash scripts/profile_named_r6.sh > gpurun_out/prof_named_r6_stdout.txt 2>&1; echo "rc $?" >> gpurun_out/prof_named_r6_stdout.txt
tail -25 gpurun_out/prof_named_r6_stdout.txt | cut -c1-400
