"""N>1 path on CPU: world-size-2 gloo processes shard the reads, receive the index by broadcast, seed their
shard (the oracle stands in for the GPU here -- this test is about the sharding / broadcast / ordering
logic, not about the kernels) and the gathered result equals the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_py as O
from common import GOLDEN, build_index, read_fastq_codes
from pymeme import shard


def test_partition_properties():
    for n in (0, 1, 511, 512, 513, 5000, 100_000):
        for w in (1, 2, 3, 8):
            parts = shard.partition(n, w)
            assert len(parts) == w and parts[0][0] == 0 and parts[-1][1] == n
            for (a, b), (c, d) in zip(parts, parts[1:]):
                assert b == c and a <= b
            assert all(lo % 512 == 0 for lo, _ in parts if lo < n)
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 2 * 512


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, prefix, fq, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        reads, off = read_fastq_codes(fq)
        n = off.shape[0] - 1
        # rank 0 owns the index image; the others receive it by broadcast
        if rank == 0:
            idx0 = O.load_index_files(prefix)
            meta = torch.tensor([idx0.text.shape[0]], dtype=torch.int64)
        else:
            meta = torch.zeros(1, dtype=torch.int64)
        dist.broadcast(meta, 0)
        m = int(meta[0])
        text = torch.from_numpy(idx0.text.copy()) if rank == 0 else torch.empty(m, dtype=torch.uint8)
        sa = torch.from_numpy(idx0.sa.view(np.int64).copy()) if rank == 0 else torch.empty(m, dtype=torch.int64)
        shard.broadcast_index([text, sa], 0)
        idx = O.Index(text.numpy(), sa.numpy().view(np.uint64))
        lo, hi = shard.partition(n, world, batch=64)[rank]
        sub_off = off[lo:hi + 1] - off[lo]
        sm, ns, hits, nh, _ = O.seed_batch(idx, reads[off[lo]:off[hi]], sub_off, smem_cap=256, hit_cap=4096, threads=1)
        dump = O.format_seed_dump(sm, ns, hits, first_id=lo)
        parts = shard.gather_in_order(dump, 0)
        t = shard.max_over_ranks(float(rank + 1))
        assert t == float(world)
        if rank == 0:
            ret.put("".join(parts))
    finally:
        dist.destroy_process_group()


def test_two_ranks_reproduce_single_process_result():
    prefix = build_index(os.path.join(GOLDEN, "g1.fa"))
    fq = os.path.join(GOLDEN, "g1_reads_150.fq")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, prefix, fq, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == open(os.path.join(GOLDEN, "g1_seeds_150.txt")).read()
