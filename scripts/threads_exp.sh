cd /root/repo
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_ext.py -x -q 2>&1 | tail -4
timeout 300 python scripts/chain_probe.py 3100 2 2>&1 | tail -2
rm -rf /dev/shm/meme_bench_* 2>/dev/null
timeout 2400 python bench.py > gpurun_out/bench_r3_a.json 2> gpurun_out/bench_r3_a.err; echo "bench rc $?"
tail -4 gpurun_out/bench_r3_a.err | cut -c1-300
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_r3_a.json') if l.startswith('{')][-1])
for k in ("value","ms_per_step"): print(k, d[k])
print("roofline", {k:d["roofline"][k] for k in ("achieved","frac","kernel_ms")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
for k in ("chain","ext","bsw"):
    print(k, json.dumps(d.get(k))[:1000])
e=d.get("e2e",{})
print("e2e", json.dumps({k:e.get(k) for k in ("value","sam_identical","speedup_wall","speedup_process")}))
print("dropin", json.dumps(e.get("dropin"))[:1500])
print("reference", json.dumps(e.get("reference"))[:600])
PY
