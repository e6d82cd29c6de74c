#!/usr/bin/env python3
"""Fixture generator (runs in the build container, where the compiled reference is; no GPU): the posing step of mate rescue --
mem_sam_pe_batch_pre + mem_matesw_batch_pre (reference src/bwamem_pair.cpp:660-716, 1060-1223) -- through oracle/ref_stage_shim.cpp
(ref_matesw_pose: the reference's own function over made-up alignment records), worker batch by worker batch, and the kswr_t records the
reference's AVX-512 kswv kernels give for the posed jobs (ref_kswv_batch).  -> tests/golden/matesw_golden.npz"""
import os, sys, zlib
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "..", "bwa-meme_amd"))
import numpy as np
import ref_py as R
from common import matesw_pose_workload

out = {}
for tag, seed, pes in (("a", 301, None), ("b", 302, [(50, 500, 0), (120, 680, 0), (100, 900, 0), (0, 0, 1)])):
    W = matesw_pose_workload(seed=seed, pes=pes)
    n = W["read_len"].shape[0]
    gars, gar_off, job_off, jl1, jl2, jx, res, crc = [], [0], [0], [], [], [], [], []
    for first in range(0, n, 512):
        count = min(512, n - first)
        g, j, ref, qer = R.matesw_pose(W["genome"], W["l_pac"], W["contig_off"], W["contig_len"], W["reads"], W["read_off"], first, count, W["regs"], W["reg_off"], W["pes"])
        gars.append(g); gar_off.append(gar_off[-1] + g.shape[0]); job_off.append(job_off[-1] + j.shape[0])
        jl1.append(j["len1"]); jl2.append(j["len2"]); jx.append(j["xtra"])
        crc.append((zlib.crc32(ref.tobytes()), zlib.crc32(qer.tobytes())))
        if j.shape[0]:
            res.append(R.kswv_batch(j, ref, qer))
    for k in ("genome", "contig_off", "contig_len", "reads", "read_off", "read_len", "regs", "reg_off", "pes"):
        out[tag + "_" + k] = W[k]
    out[tag + "_l_pac"] = np.int64(W["l_pac"])
    out[tag + "_gar"] = np.concatenate(gars); out[tag + "_gar_off"] = np.array(gar_off, np.int64); out[tag + "_job_off"] = np.array(job_off, np.int64)
    out[tag + "_len1"] = np.concatenate(jl1); out[tag + "_len2"] = np.concatenate(jl2); out[tag + "_xtra"] = np.concatenate(jx)
    out[tag + "_kswr"] = np.concatenate(res)
    out[tag + "_seq_crc"] = np.array(crc, np.uint32)
    print(tag, "pairs", n // 2, "jobs", out[tag + "_len1"].shape[0], "gar", out[tag + "_gar"].shape[0])
np.savez_compressed(os.path.join(HERE, "matesw_golden.npz"), **out)
