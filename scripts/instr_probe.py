import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))
import numpy as np, torch
from pymeme import hipapi, hostapi, synth, workload
mbp = float(sys.argv[1]); n = 1000000
g = synth.make_genome(int(mbp * 1e6), seed=11)
text, sa = hostapi.build_sa(g); l1, l2 = hostapi.train_prmi(text, sa)
ctx = hipapi.Context(0)
pp = np.zeros((sa.shape[0], 5), np.uint8); pp[:, :4] = (sa >> np.uint64(8)).astype('<u4').view(np.uint8).reshape(-1, 4); pp[:, 4] = (sa & np.uint64(255)).astype(np.uint8)
ctx.load_index_host(pp.reshape(-1), text, l1, l2)
reads = workload.make_reads_fast(g, n, 150, seed=12)
d_reads = torch.from_numpy(reads.reshape(-1)).cuda(); d_off = torch.arange(0, (n + 1) * 150, 150, dtype=torch.int64, device="cuda")
for rounds in (1, 2, 3):
    res = ctx.seed_batch_device(d_reads.data_ptr(), d_off.data_ptr(), n, n * 150, hipapi.default_seed_opt(rounds=rounds))
    w = ctx.timings().seed_windows
    print("rounds", rounds, "searches/read %.1f" % (res.searches / n), "windows/search %.3f" % ((w & 0xfffff) / res.searches), "relocations/search %.3f" % (((w >> 20) & 0xfffff) / res.searches), "edge calls/search %.3f" % ((w >> 40) / res.searches))
