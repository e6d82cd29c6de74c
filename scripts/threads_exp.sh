cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_chain.py tests/test_gpu_ext.py -x -q -m gpu 2>&1 | tail -3
for q in 8 4; do
GPU_MAX_HW_QUEUES=$q rocprofv3 --kernel-trace --stats -d gpurun_out/chain_trace -o c -- python scripts/chain_probe.py 3100 2 > gpurun_out/chain_trace.log 2>&1
echo "== GPU_MAX_HW_QUEUES=$q"; grep "chain kernels" gpurun_out/chain_trace.log
python scripts/rocpd_timeline.py gpurun_out/chain_trace/c_results.db k_chain 10
rm -rf gpurun_out/chain_trace
done
