#!/bin/bash
# Round 5, GPU call Q.  SURVEY 8 (f)2: kernel trace of the CIGAR stage in both read classes (which kernel the 62 ms of the 250-bp class belong to).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q; mkdir -p $O
export MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=1 MEME_BENCH_C4=1 MEME_BENCH_C4_E2E=0 MEME_BENCH_PMC=0 MEME_BENCH_PARITY_READS=50000
export ROCPD_KERNELS=k_gcig,k_md,k_cjob ROCPD_ROWS=40
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/p1.err
python scripts/rocpd_summary.py $O/trace/t_results.db > $O/trace.md 2>&1
python scripts/rocpd_timeline.py $O/trace/t_results.db k_gcig 40 > $O/timeline.txt 2>&1; rm -rf $O/trace
grep -h "k_gcig\|k_md\|k_cjob" $O/trace.md | head -20; tail -30 $O/timeline.txt | cut -c1-160
