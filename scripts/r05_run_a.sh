#!/bin/bash
# Round 5, GPU call A.  SURVEY 8 rows served: (f)2 (bwa_gen_cigar2 whole on the device: parity tests), (f)4 (SAM-phase split of the bound
# aligner at GRCh38 size: bwa-meme_dropin_prof), B8 (census of exact-prefix extension jobs in the ext leg), e2e after the change.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_gcig.py tests/test_gpu_sam_e2e.py tests/test_gpu_ext.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=1 MEME_BENCH_E2E_PAIRS=2000000 MEME_BENCH_E2E_SKIP_REF=1 \
MEME_BENCH_E2E_DROPIN_EXE=bwa-meme_dropin,bwa-meme_dropin_prof MEME_BENCH_E2E_STDERR=$O/e2e MEME_BENCH_PARITY_READS=50000 \
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
echo "bench rc $?" >> $O/bench.err
tail -3 $O/pytest.log; tail -5 $O/bench.err
