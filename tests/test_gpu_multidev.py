"""Multi-GPU as a library feature (C ABI meme_index_replicate) and as one process per GPU (RCCL broadcast of the raw index
images + the staging kernels on every rank).  The two-device cases skip on a one-GPU box; the same-device case always runs."""
import os
import socket

import numpy as np
import pytest

import oracle_py as O
from common import GOLDEN, build_index, read_fastq_codes
from pymeme import hipapi, shard

pytestmark = pytest.mark.gpu


def _dump(ctx, reads, off):
    smems, smem_off, hits, hit_off = ctx.seed_batch(reads, off)
    slots, counts, hl = hipapi.smems_to_slots(smems, smem_off, hits, hit_off)
    return O.format_seed_dump(slots, counts, hl)


def _n_devices():
    return hipapi.lib().meme_device_count()


def test_replicate_on_the_same_device_shares_the_index():
    prefix = build_index(os.path.join(GOLDEN, "g1.fa"))
    reads, off = read_fastq_codes(os.path.join(GOLDEN, "g1_reads_150.fq"))
    want = open(os.path.join(GOLDEN, "g1_seeds_150.txt")).read()
    a = hipapi.Context(0)
    b = hipapi.Context(0)
    try:
        a.load_index_files(prefix)
        b.replicate_index_from(a)
        assert _dump(b, reads, off) == want
        assert b.describe_index().d_sa_ent == a.describe_index().d_sa_ent
    finally:
        b.close()
        a.close()


@pytest.mark.skipif(_n_devices() < 2, reason="needs two GPUs")
def test_replicate_to_a_second_device():
    prefix = build_index(os.path.join(GOLDEN, "g1.fa"))
    a = hipapi.Context(0)
    b = hipapi.Context(1)
    try:
        a.load_index_files(prefix)
        b.replicate_index_from(a)
        assert b.describe_index().d_sa_ent != a.describe_index().d_sa_ent
        for length in (150, 250):
            reads, off = read_fastq_codes(os.path.join(GOLDEN, "g1_reads_%d.fq" % length))
            want = open(os.path.join(GOLDEN, "g1_seeds_%d.txt" % length)).read()
            assert _dump(b, reads, off) == want
    finally:
        b.close()
        a.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, prefix, fq, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        reads, off = read_fastq_codes(fq)
        n_reads = off.shape[0] - 1
        meta = torch.zeros(3, dtype=torch.int64, device=dev)
        if rank == 0:
            text = np.fromfile(prefix + ".0123", dtype=np.uint8)
            pos5 = np.fromfile(prefix + ".pos_packed", dtype=np.uint8)
            l1, l2 = O.load_prmi_files(prefix)
            meta[0], meta[1], meta[2] = text.shape[0], l2.shape[0], l1.shape[0]
        dist.broadcast(meta, 0)
        n, n_l2, n_l1 = (int(x) for x in meta)
        L = hipapi.lib()
        d_text = torch.empty(n, dtype=torch.uint8, device=dev)
        d_pos5 = torch.zeros(L.meme_index_pos5_bytes(n), dtype=torch.uint8, device=dev)
        d_l2 = torch.empty(n_l2 * 24, dtype=torch.uint8, device=dev)
        d_l1 = torch.zeros(max(n_l1, 1) * 24, dtype=torch.uint8, device=dev)
        if rank == 0:
            d_text.copy_(torch.from_numpy(text))
            d_pos5[:pos5.shape[0]].copy_(torch.from_numpy(pos5))
            d_l2.copy_(torch.from_numpy(l2.view(np.uint8).reshape(-1)))
            if n_l1:
                d_l1[:n_l1 * 24].copy_(torch.from_numpy(np.ascontiguousarray(l1).view(np.uint8).reshape(-1)))
        shard.broadcast_index([d_text, d_pos5, d_l2, d_l1], 0)          # RCCL over xGMI
        ctx = hipapi.Context(rank)
        keep = hipapi.stage_index_torch(ctx, n, d_text, d_pos5, d_l2, n_l2, d_l1, n_l1)
        lo, hi = shard.partition(n_reads, world, batch=64)[rank]
        sub_off = off[lo:hi + 1] - off[lo]
        smems, so, hits, ho = ctx.seed_batch(reads[off[lo]:off[hi]], sub_off)
        slots, counts, hl = hipapi.smems_to_slots(smems, so, hits, ho)
        parts = shard.gather_in_order(O.format_seed_dump(slots, counts, hl, first_id=lo), 0)
        ctx.close()
        del keep
        if rank == 0:
            ret.put("".join(parts))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(_n_devices() < 2, reason="needs two GPUs")
def test_two_ranks_over_rccl_reproduce_the_golden_dump():
    import torch.multiprocessing as mp
    prefix = build_index(os.path.join(GOLDEN, "g1.fa"))
    fq = os.path.join(GOLDEN, "g1_reads_150.fq")
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank, args=(r, 2, port, prefix, fq, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == open(os.path.join(GOLDEN, "g1_seeds_150.txt")).read()
