#!/bin/bash
# Round 6, GPU call N10.  bench.py's counter passes report twice the FETCH_SIZE when the child takes the parent's saved reads instead of sampling them again -- at the
# named configuration only (calls R2-R4; not on a 512-Mbp probe, call N9).  The two children side by side, seeding only, 2 launches each under --pmc FETCH_SIZE: per-kernel
# sums and the bench line's own work counters (SMEMs, hits, searches, windows per read) -- the same input or not?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n10; mkdir -p $O; RAW=/tmp/r06n10_raw; mkdir -p $RAW
export MEME_BENCH_PMC=0 MEME_BENCH_CPU=0 MEME_BENCH_E2E=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_RD=0 MEME_BENCH_PARITY_READS=0
export ROCPD_KERNELS=k_seed,k_reseed ROCPD_ROWS=40
# the reads file exactly as bench.py's parent writes it: one plain seeding-only run that saves its batch (MEME_BENCH_KEEP_READS names the copy)
MEME_BENCH_KEEP_READS=/dev/shm/n10_reads.npy timeout 300 python bench.py --steps 1 --warmup 0 > $O/parent.json 2> $O/parent.err
ls -la /dev/shm/n10_reads.npy >> $O/parent.err 2>&1
for v in sampled cached; do
  rm -rf $RAW/x
  if [ $v = cached ]; then export MEME_BENCH_READS_FILE=/dev/shm/n10_reads.npy; else unset MEME_BENCH_READS_FILE; fi
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $RAW/x -o t -- python bench.py --steps 2 --warmup 0 > $O/child_$v.json 2> $O/child_$v.err
  python scripts/rocpd_summary.py $RAW/x/t_results.db > $O/pmc_$v.md 2>&1
  rm -rf $RAW/x
  echo "== $v"; grep -h "FETCH_SIZE" $O/pmc_$v.md | cut -c1-160
  python - <<PY
import json
d = json.loads(open("$O/child_$v.json").read().strip().split("\n")[-1])
c = d["config"]
print("$v", {k: c.get(k) for k in ("smems_per_read", "hits_per_read", "searches_per_read", "windows_per_search")}, "stage ms", d["roofline"]["kernel_ms"])
PY
done
rm -f /dev/shm/n10_reads.npy
