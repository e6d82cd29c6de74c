cd /root/repo
timeout 600 python -m pytest tests/test_gpu_gcig.py -x -q 2>&1 | tail -15
for cfg in "64 0" "64 1" "96 1" "40 1"; do
  set -- $cfg
  E2E_SKIP_REF=1 MEME_DROPIN_IO=$2 timeout 900 python scripts/e2e_bench.py 128 2 $1 > gpurun_out/e2e_threads_$1_io$2.log 2>&1
  echo "threads $1 io $2"; grep -E "dropin device rc|WORKER_SAM|MEM_PROCESS_SEQ|Reading IO time \(reads\)|main_mem" gpurun_out/e2e_threads_$1_io$2.log | cut -c1-160
done
