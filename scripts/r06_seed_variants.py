#!/usr/bin/env python3
"""Round 6, the three bounded experiments on the SA-search stage (reference src/LearnedIndex_seeding.cpp:2131-2664).
ONE index at the named configuration (GRCh38-sized synthetic genome, built on the device once), then every library variant and every model size on
it, kernel times by HIP events (meme_get_timings), windows per search from the kernel's own counter:

  python scripts/r06_seed_variants.py [Mbp] [Mreads] "name=lib.so,..." "bits,bits,..." [out.json]

Variants are whole builds of libmeme_hip.so (scripts/build_variants.sh name="-DSEED_ALIGN_WIN=1" ...); the model-size sweep retrains the P-RMI on
the device with 2^bits leaves (meme_prmi_train_device) and re-attaches.  Under `rocprofv3 --pmc ...` the same script gives lines per search."""
import ctypes as C, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))
import numpy as np, torch
from pymeme import hipapi, synth, workload

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100
mreads = float(sys.argv[2]) if len(sys.argv) > 2 else 10
variants = [v.split("=", 1) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 and sys.argv[3] else [["head", os.path.join(REPO, "bwa-meme_amd", "libmeme_hip.so")]]
bits_list = [int(b) for b in sys.argv[4].split(",")] if len(sys.argv) > 4 and sys.argv[4] else [0]
out_json = sys.argv[5] if len(sys.argv) > 5 else None
RL = int(os.environ.get("PROBE_READ_LEN", "150")); SUB = float(os.environ.get("PROBE_SUB_RATE", "0.01")); INDEL = float(os.environ.get("PROBE_INDEL_RATE", "0"))
STEPS = int(os.environ.get("PROBE_STEPS", "4"))
log = lambda s: print("[variants]", s, flush=True)

def use_lib(path):
    hipapi._lib = None
    hipapi.LIB_PATH = path if os.path.isabs(path) else os.path.join(REPO, "bwa-meme_amd", path)
    return hipapi.lib()

l_pac = int(mbp * 1e6) & ~1
n = 2 * l_pac
t0 = time.time()
g = synth.make_genome(l_pac, seed=11)
use_lib(variants[0][1])
ctx = hipapi.Context(0)
text = hipapi.fwd_rc_text(g)
d_text, d_sa = hipapi.build_sa_device(ctx, text)
d_pos5 = hipapi.pos5_from_sa_torch(ctx, d_sa, n)
del d_sa, text
torch.cuda.empty_cache()
d_pac, d_ent = hipapi.stage_entries_torch(ctx, n, d_text, d_pos5)
del d_text, d_pos5
torch.cuda.empty_cache()
auto_bits = 28 if 8.0 * n + 8 > 8.0e9 else 26 if 8.0 * n + 8 > 1.0e9 else 24
log("genome %.0f Mbp + entries on the device in %.1f s" % (l_pac / 1e6, time.time() - t0))
nreads = int(mreads * 1e6)
if INDEL > 0:
    reads = workload.make_reads_fast(g, nreads, RL, seed=1000, sub_rate=SUB, indel_rate=INDEL) if "indel_rate" in workload.make_reads_fast.__code__.co_varnames else workload.make_reads_fast(g, nreads, RL, seed=1000, sub_rate=SUB)
else:
    reads = workload.make_reads_fast(g, nreads, RL, seed=1000, sub_rate=SUB)
d_reads = torch.from_numpy(reads.reshape(-1)).cuda()
d_off = torch.arange(0, (nreads + 1) * RL, RL, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
models = {}
for b in bits_list:
    bb = b if b > 0 else auto_bits
    models[bb] = hipapi.train_prmi_device(ctx, d_ent, n, bb)
    log("model with 2^%d leaves: %d partial records" % (bb, models[bb][3]))
ctx.close()
rows = []
ref_sig = None
for name, path in variants:
    use_lib(path)
    for bb, (d_l2, n_l2, d_l1, n_l1) in models.items():
        c = hipapi.Context(0)
        keep = hipapi.attach_index_torch(c, n, d_pac, d_ent, d_l2, n_l2, d_l1, n_l1)
        for key in [k for k in os.environ.get("PROBE_TUNING", "").split(",") if k]:
            c.set_tuning(key.split("=")[0], int(key.split("=")[1]))
        ms = []
        for it in range(STEPS + 1):
            res = c.seed_batch_device(d_reads.data_ptr(), d_off.data_ptr(), nreads, nreads * RL, hipapi.default_seed_opt(rounds=3))
            tm = c.timings()
            if it:
                ms.append((tm.seed_kernel_ms, tm.seed_reseed_ms, tm.seed_gather_ms + tm.seed_pack_ms))
        sig = (res.total_smems, res.total_hits)
        if ref_sig is None:
            ref_sig = sig
        row = {"variant": name, "bits": bb, "stage_ms": float(np.mean([m[0] for m in ms])), "stage_ms_min": float(np.min([m[0] for m in ms])), "reseed_kernels_ms": float(np.mean([m[1] for m in ms])),
               "pack_gather_ms": float(np.mean([m[2] for m in ms])), "M_reads_per_s_stage": nreads / float(np.mean([m[0] for m in ms])) / 1e3,
               "searches_per_read": res.searches / nreads, "windows_per_read": tm.seed_windows / nreads, "windows_per_search": tm.seed_windows / max(res.searches, 1),
               "smems": int(res.total_smems), "hits": int(res.total_hits), "same_totals_as_first": sig == ref_sig}
        rows.append(row)
        log(json.dumps(row))
        c.close()
        del keep
        torch.cuda.empty_cache()
if out_json:
    json.dump({"genome_bp": l_pac, "reads": nreads, "read_len": RL, "sub_rate": SUB, "rows": rows}, open(out_json, "w"), indent=1)
