#!/bin/bash
# Round 5, GPU call X.  SURVEY 8 (f)2: the counters profiles/r05_gcig.md names as next for k_gcig (250-bp class) -- LDS waits / conflicts, instruction fetch,
# scalar cache -- taken only from what `rocprofv3 --list-avail` offers on this box.  Probe mode (the ext leg reports no value); raw output under /tmp.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05x; mkdir -p $O; RAW=/tmp/r05x_raw
(timeout 120 rocprofv3 --list-avail 2>&1 || timeout 120 rocprofv3 -L 2>&1) > /tmp/avail.txt
grep -oE "\b(SQ|SQC|TCP|TA|TD)_[A-Z0-9_]+" /tmp/avail.txt | sort -u > $O/counters_avail.txt
echo "counters offered (SQ/SQC/TCP/TA/TD): $(wc -l < $O/counters_avail.txt)"
WANT1="SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL"
WANT2="SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM"
WANT3="SQ_WAVES SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_LDS"
export MEME_BENCH_MBP=256 MEME_BENCH_READS=2000000 MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=1 MEME_BENCH_C4_E2E=0 MEME_BENCH_PMC=0 MEME_BENCH_PARITY_READS=20000 MEME_BENCH_EXT_CHECK=0
export ROCPD_KERNELS=k_gcig ROCPD_ROWS=20
n=0
for W in "$WANT1" "$WANT2" "$WANT3"; do
  n=$((n+1)); HAVE=""; MISS=""
  for c in $W; do if grep -qx "$c" $O/counters_avail.txt; then HAVE="$HAVE $c"; else MISS="$MISS $c"; fi; done
  echo "== pass $n: collecting$HAVE ; not offered here:$MISS"
  [ -z "$HAVE" ] && continue
  rm -rf $RAW
  timeout 240 rocprofv3 --pmc $HAVE -d $RAW -o t -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $O/pass$n.err
  if [ -f $RAW/t_results.db ]; then python scripts/rocpd_summary.py $RAW/t_results.db > $O/pass$n.md 2>&1; grep -h "k_gcig(" $O/pass$n.md | cut -c1-150; else echo "pass $n produced no database:"; tail -4 $O/pass$n.err | cut -c1-200; fi
  rm -rf $RAW
done
du -sh gpurun_out | tail -1
