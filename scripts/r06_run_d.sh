#!/bin/bash
# Round 6, GPU call D.  SURVEY 8(a) S-rows: lines per search of the seeding variants (FETCH_SIZE pass); 8(d)/(f): the repeat-dense
# workload (tests + bench leg); 8(e): device-stage floor per slice of a chunk at the named configuration.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_repeat_dense.py -x -q -m gpu -s > $O/pytest_rd.log 2>&1; echo "pytest rc $?" >> $O/pytest_rd.log
grep -E "passed|failed|rc |repeat-dense:|Error|assert" $O/pytest_rd.log | tail -8
( cd /tmp && PROBE_STEPS=2 timeout 900 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_var -o pmc -- python $GRAFT_REPO_ROOT/scripts/r06_seed_variants.py 3100 10 "head=libmeme_hip.so,align=libmeme_hip_align.so" "28,22" > $GRAFT_REPO_ROOT/$O/pmc_variants.log 2>&1 )
python - <<'PY' > $O/pmc_variants_summary.txt 2>&1
import glob, sqlite3
dbs = glob.glob("/tmp/pmc_var/**/*results.db", recursive=True)
cur = sqlite3.connect(dbs[0]).cursor()
rows = cur.execute("select dispatch_id, kernel_name, value from counters_collection where counter_name = 'FETCH_SIZE' order by dispatch_id").fetchall()
seed = [(d, k, v) for d, k, v in rows if "k_seed" in k]
resd = [(d, k, v) for d, k, v in rows if "k_reseed" in k]
print("k_seed launches:", len(seed))
# 3 launches per (variant, bits) combination, in order head/28, head/22, align/28, align/22
names = ["head b28", "head b22", "align b28", "align b22"]
per = len(seed) // len(names) if seed else 0
for i, nm in enumerate(names):
    part = seed[i * per:(i + 1) * per]
    if not part: continue
    kb = sum(v for _, _, v in part) / len(part)
    print("%s: k_seed FETCH_SIZE %.0f KB per launch -> %.1f GB fetched (x2 rule) = %.3f G lines" % (nm, kb, 2 * kb * 1024 / 1e9, 2 * kb * 1024 / 128 / 1e9))
PY
cat $O/pmc_variants_summary.txt; grep "variants\]" $O/pmc_variants.log | cut -c1-260 | tail -4
rm -rf /tmp/pmc_var
MEME_BENCH_CPU=0 MEME_BENCH_BSW=0 MEME_BENCH_KSWV=0 MEME_BENCH_CHAIN=0 MEME_BENCH_EXT=0 MEME_BENCH_C4=0 MEME_BENCH_PMC=0 MEME_BENCH_E2E_SKIP_REF=1 MEME_BENCH_E2E_PAIRS=4000000 \
MEME_BENCH_PARITY_READS=50000 MEME_BENCH_E2E_STDERR=$O/e2e timeout 1700 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
grep -E "repeat|e2e:|bench rc|failed" $O/bench.err | cut -c1-300
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06d/bench.json").read().strip().split("\n")[-1])
print(json.dumps(d.get("repeat_dense"), indent=0)[:3000])
print(json.dumps(d.get("e2e", {}).get("dropin", {}).get("bound"), indent=0)[:2500])
PY
du -sh gpurun_out | tail -1
