#!/bin/bash
# Round 6, GPU call K.  SURVEY 8(d): the whole default bench line on the code so far (every leg, the new repeat_dense leg and the slice floor included), with its timeline; + the mate tests.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_mate.py -q -m gpu > $O/pytest_mate.log 2>&1; echo "pytest rc $?" >> $O/pytest_mate.log; tail -3 $O/pytest_mate.log
MEME_BENCH_E2E_STDERR=$O/e2e timeout 1790 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
grep -E "^\[bench" $O/bench.err | cut -c1-230
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06k/bench.json").read().strip().split("\n")[-1])
r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", r["frac"], "traffic/alg", r.get("traffic_over_algorithmic"), "rand-line frac", r.get("frac_of_random_line_ceiling"))
print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "kind")})
for k in ("bsw", "kswv", "chain"):
    print(k, {x: d[k][x] for x in list(d[k])[:8]} if d.get(k) else None)
e = d["ext"]; print("ext in_rounds", e["in_rounds"], "cigar", e["cigar"]["kernel_ms"], e["cigar"]["matches_oracle"])
c4 = d["config4_class"]; print("c4 seeding", c4["seeding"]["value"], c4["seeding"]["roofline"]["frac"], "chain", c4["chain"]["kernel_ms"], "ext", c4["ext"]["in_rounds"], "cigar", c4["ext"]["cigar"]["kernel_ms"], "e2e", {k: c4.get("e2e", {}).get(k) for k in ("sam_identical", "speedup_process")})
rd = d.get("repeat_dense"); print("rd", rd.get("all_checks_true") if rd else None, rd["seeding"]["value"] if rd and "seeding" in rd else rd)
e2 = d["e2e"]; print("e2e", e2.get("sam_identical"), e2["dropin"]["process_s"], e2["dropin"]["process_cpu_s"], e2["reference"]["process_s"] if e2.get("reference") else None, e2["dropin"]["wall_s"], e2["dropin"].get("backend"))
PY
rm -f $O/e2e/bwa-meme_mode3*.stderr
