#!/bin/bash
# Round 5, GPU call R.  SURVEY 8 (f)2: counters of k_gcig (what a CIGAR job's 130 ns are: issue, waits, occupancy) on a 256 Mbp probe genome, both
# read classes; the LDS kept per job for the backtrack window swept (tuning gcig_zcap).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05r; mkdir -p $O
export MEME_BENCH_MBP=256 MEME_BENCH_READS=2000000 MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=1 MEME_BENCH_C4=1 MEME_BENCH_C4_E2E=0 MEME_BENCH_PMC=0 MEME_BENCH_PARITY_READS=20000
for z in 8192 4096 2048 0; do
  MEME_TUNING="gcig_zcap=$z" python bench.py --steps 2 --warmup 1 > $O/bench_z$z.json 2> $O/bench_z$z.err
  python - $z <<'PY'
import json,sys
z=sys.argv[1]
d=json.loads(open('gpurun_out/r05r/bench_z%s.json'%z).read().strip().splitlines()[-1])
print('zcap',z,'150bp cigar ms',d['ext']['cigar']['kernel_ms'],'ok',d['ext']['cigar']['matches_oracle'],'| 250bp cigar ms',d['config4_class']['ext']['cigar']['kernel_ms'],'ok',d['config4_class']['ext']['cigar']['matches_oracle'])
PY
done
export ROCPD_KERNELS=k_gcig ROCPD_ROWS=40
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc1 -o t -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $O/p1.err
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES -d $O/pmc2 -o t -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $O/p2.err
rocprofv3 --pmc WRITE_SIZE FETCH_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/pmc3 -o t -- python bench.py --steps 2 --warmup 1 > /dev/null 2> $O/p3.err
for d in pmc1 pmc2 pmc3; do python scripts/rocpd_summary.py $O/$d/t_results.db > $O/$d.md 2>&1; rm -rf $O/$d; done
grep -h "k_gcig(" $O/pmc1.md $O/pmc2.md $O/pmc3.md | cut -c1-200
