cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/chain_trace -o c -- python scripts/chain_probe.py 3100 2 > gpurun_out/chain_trace.log 2>&1
ROCPD_ROWS=60 python scripts/rocpd_summary.py gpurun_out/chain_trace/c_results.db > gpurun_out/chain_trace.md 2>&1; rm -rf gpurun_out/chain_trace
grep -E "k_chain|k_scan" gpurun_out/chain_trace.md
