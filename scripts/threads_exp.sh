cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_sam_e2e.py -x -q 2>&1 | tail -4
for io in 1 0 1; do
E2E_SKIP_REF=1 MEME_DROPIN_IO=$io timeout 900 python scripts/e2e_bench.py 128 2 64 > gpurun_out/e2e_150_io$io.log 2>&1
echo "150bp io $io"; grep -E "dropin device rc|WORKER_SAM|MEM_PROCESS_SEQ|main_mem|Reading IO time \(reads\)|Writing IO" gpurun_out/e2e_150_io$io.log | cut -c1-200
done
