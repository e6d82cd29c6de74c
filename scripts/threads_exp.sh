cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/chain_trace -o c -- python scripts/chain_probe.py 3100 2 > gpurun_out/chain_trace.log 2>&1
grep "chain kernels" gpurun_out/chain_trace.log
python scripts/rocpd_timeline.py gpurun_out/chain_trace/c_results.db k_chain 11
rm -rf gpurun_out/chain_trace
