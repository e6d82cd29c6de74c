cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_sam_e2e.py -x -q 2>&1 | tail -15
for cg in 1 0; do
  E2E_READ_LEN=250 E2E_SUB=0.05 E2E_SKIP_REF=1 MEME_DROPIN_CIGAR=$cg timeout 900 python scripts/e2e_bench.py 128 1 64 > gpurun_out/e2e_250_cigar$cg.log 2>&1
  echo "250bp/5% cigar stage $cg"; grep -E "dropin device rc|WORKER_SAM|MEM_PROCESS_SEQ|Reading IO time \(reads\)|main_mem|CIGAR stage|SAM md5" gpurun_out/e2e_250_cigar$cg.log | cut -c1-420
done
