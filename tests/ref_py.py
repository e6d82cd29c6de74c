"""ctypes / subprocess access to the *compiled reference* under oracle/_ref (TEST INFRASTRUCTURE ONLY).

oracle/_ref is built from /root/reference by oracle/Makefile.ref in the build container; the binaries
travel to the GPU box, the sources do not.  Everything here degrades to "not available" when the
directory is missing or the host CPU cannot run the binaries (they are compiled for AVX-512 when the
build host has it).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from oracle_py import SEQPAIR_DTYPE, OrcBswParams, REPO

REF_DIR = os.path.join(REPO, "oracle", "_ref")
_BSW = None


def have(name: str) -> bool:
    return os.path.exists(os.path.join(REF_DIR, name))


def cpu_can_run() -> bool:
    """The reference objects are built with the SIMD level recorded in oracle/_ref/SIMD."""
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    want = "avx512bw"
    p = os.path.join(REF_DIR, "SIMD")
    if os.path.exists(p):
        want = open(p).read().strip().lstrip("-m") or want
    return (want in flags) or want.startswith("sse")


def bsw_lib():
    global _BSW
    if _BSW is None:
        _BSW = C.CDLL(os.path.join(REF_DIR, "libbsw_ref.so"))
        assert _BSW.ref_sizeof_seqpair() == SEQPAIR_DTYPE.itemsize
    return _BSW


def bsw_run(kind: int, pairs: np.ndarray, ref: np.ndarray, qer: np.ndarray, w: int, params: OrcBswParams):
    """kind 0 = scalarBandedSWAWrapper, 16 = getScores16, 8 = getScores8 (reference src/bandedSWA.h).
    Returns a copy of `pairs` with the six outputs filled by the reference."""
    n = pairs.shape[0]
    buf = np.zeros(n + 128, dtype=SEQPAIR_DTYPE)  # the SIMD wrappers pad past numPairs
    buf[:n] = pairs
    # the SIMD wrappers read sequence bytes of the padding pairs too -> keep slack in the buffers
    ref = np.concatenate([np.ascontiguousarray(ref, dtype=np.uint8), np.zeros(1 << 16, np.uint8)])
    qer = np.concatenate([np.ascontiguousarray(qer, dtype=np.uint8), np.zeros(1 << 16, np.uint8)])
    rc = bsw_lib().ref_bsw_run(C.c_int(kind), C.c_void_p(buf.ctypes.data), C.c_void_p(ref.ctypes.data),
                               C.c_void_p(qer.ctypes.data), C.c_int32(n), C.c_int32(w), C.byref(params))
    assert rc == 0
    return buf[:n].copy()


def run_seed_dump(prefix: str, fastq: str, mode: int = 3, steps: int = 4, threads: int = 1, timeout=600) -> str:
    exe = os.path.join(REF_DIR, "learned_seeding_mode%d" % mode)
    out = subprocess.run([exe, prefix, fastq, "1000", str(threads), str(steps)], capture_output=True,
                         timeout=timeout, check=True)
    return out.stdout.decode()


# ---- the stages between the two kernels: the reference's own functions behind oracle/ref_stage_shim.cpp -----------------------
_STAGE = None


def stage_lib():
    global _STAGE
    if _STAGE is None:
        _STAGE = C.CDLL(os.path.join(REF_DIR, "libstage_ref.so"))
    return _STAGE


def chain_read(smems, hits, read_len, contig_off, contig_len, contig_alt, opt, chain_cap=8192, seed_cap=1 << 17):
    """mem_chain_Learned + mem_chain_flt of the compiled reference for one read; same return shape as oracle_py.chain_read."""
    from oracle_py import MEM_TL_DTYPE, ORC_CHAIN_DTYPE, ORC_CSEED_DTYPE
    L = stage_lib()
    L.ref_chain_read.restype = C.c_int
    smems = np.ascontiguousarray(smems, dtype=MEM_TL_DTYPE)
    hits = np.ascontiguousarray(hits, dtype=np.uint64)
    contig_off = np.ascontiguousarray(contig_off, dtype=np.int64)
    contig_len = np.ascontiguousarray(contig_len, dtype=np.int32)
    contig_alt = np.ascontiguousarray(contig_alt, dtype=np.uint8)
    out = np.zeros(chain_cap, ORC_CHAIN_DTYPE)
    sd = np.zeros(seed_cap, ORC_CSEED_DTYPE)
    tree, frac = C.c_int(0), C.c_uint32(0)
    rc = L.ref_chain_read(C.c_void_p(smems.ctypes.data), C.c_int(smems.shape[0]), C.c_void_p(hits.ctypes.data), C.c_int64(hits.shape[0]),
                          C.c_int(int(read_len)), C.c_void_p(contig_off.ctypes.data), C.c_void_p(contig_len.ctypes.data),
                          C.c_void_p(contig_alt.ctypes.data), C.c_int(contig_off.shape[0]), C.byref(opt), C.c_void_p(out.ctypes.data),
                          C.c_int(chain_cap), C.c_void_p(sd.ctypes.data), C.c_int(seed_cap), C.byref(tree), C.byref(frac))
    n = max(rc, 0)
    ns = int(out["n_seeds"][:n].sum())
    return rc, out[:n], sd[:ns], tree.value, np.uint32(frac.value).view(np.float32)


def flt_chained_seeds(reads, read_off, chain_off, chains, seed_off, seeds, text, l_pac, contig_off, contig_len, contig_alt, opt, min_chain_weight=0):
    """mem_flt_chained_seeds of the compiled reference on chains the caller brings: (chains, seed_off, seeds, score) after the filter."""
    from oracle_py import ORC_CHAIN_DTYPE, ORC_CSEED_DTYPE, _repack_filtered
    L = stage_lib()
    L.ref_flt_chained_seeds.restype = C.c_int
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    read_off = np.ascontiguousarray(read_off, dtype=np.int64)
    chain_off = np.ascontiguousarray(chain_off, dtype=np.int64)
    chains = np.array(chains, dtype=ORC_CHAIN_DTYPE)
    seed_off = np.ascontiguousarray(seed_off, dtype=np.int64)
    seeds = np.array(seeds, dtype=ORC_CSEED_DTYPE)
    text = np.ascontiguousarray(text, dtype=np.uint8)
    contig_off = np.ascontiguousarray(contig_off, dtype=np.int64)
    contig_len = np.ascontiguousarray(contig_len, dtype=np.int32)
    contig_alt = np.ascontiguousarray(contig_alt, dtype=np.uint8)
    n = read_off.shape[0] - 1
    score = np.zeros(seeds.shape[0], np.int32)
    kept = np.zeros(n, np.int64)
    p = lambda a: C.c_void_p(a.ctypes.data)
    rc = L.ref_flt_chained_seeds(p(reads), p(read_off), C.c_int64(n), p(chain_off), p(chains), p(seed_off), p(seeds), p(score), p(text), p(contig_off), p(contig_len),
                                 p(contig_alt), C.c_int(contig_off.shape[0]), C.c_int64(int(l_pac)), C.byref(opt), C.c_int(int(min_chain_weight)), p(kept))
    assert rc == 0, rc
    return _repack_filtered(chains, seed_off, seeds, score, kept)


def extend_reads(reads, read_off, chain_off, chains, seed_off, seeds, frac_rep, text, l_pac, contig_off, contig_len, contig_alt, opt, seed_score=None):
    """mem_chain2aln_across_reads_V2 of the compiled reference (its own BandedPairWiseSW kernels) on chains the caller brings; same
    layout as oracle_py.extend_batch."""
    from oracle_py import ORC_ALNREG_DTYPE, ORC_CHAIN_DTYPE, ORC_CSEED_DTYPE
    L = stage_lib()
    L.ref_extend_reads.restype = C.c_int
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    read_off = np.ascontiguousarray(read_off, dtype=np.int64)
    chain_off = np.ascontiguousarray(chain_off, dtype=np.int64)
    chains = np.ascontiguousarray(chains, dtype=ORC_CHAIN_DTYPE)
    seed_off = np.ascontiguousarray(seed_off, dtype=np.int64)
    seeds = np.ascontiguousarray(seeds, dtype=ORC_CSEED_DTYPE)
    frac_bits = np.ascontiguousarray(frac_rep, dtype=np.float32).view(np.uint32)
    text = np.ascontiguousarray(text, dtype=np.uint8)
    contig_off = np.ascontiguousarray(contig_off, dtype=np.int64)
    contig_len = np.ascontiguousarray(contig_len, dtype=np.int32)
    contig_alt = np.ascontiguousarray(contig_alt, dtype=np.uint8)
    out = np.zeros(int(seed_off[-1]), ORC_ALNREG_DTYPE)
    p = lambda a: C.c_void_p(a.ctypes.data)
    L.ref_extend_reads_scored.restype = C.c_int
    if seed_score is not None:
        seed_score = np.ascontiguousarray(seed_score, dtype=np.int32)
    rc = L.ref_extend_reads_scored(p(reads), p(read_off), C.c_int64(read_off.shape[0] - 1), p(chain_off), p(chains), p(seed_off), p(seeds),
                            p(seed_score) if seed_score is not None else C.c_void_p(0), p(frac_bits), p(text),
                            p(contig_off), p(contig_len), p(contig_alt), C.c_int(contig_off.shape[0]), C.c_int64(int(l_pac)), C.byref(opt), p(out))
    assert rc == 0, rc
    return out


def ksw_global2(query, target, w, a=1, b=4, o_del=6, e_del=1, o_ins=6, e_ins=1):
    """ksw_global2 of the compiled reference: (score, cigar)."""
    L = stage_lib()
    L.ref_ksw_global2.restype = C.c_int
    query = np.ascontiguousarray(query, dtype=np.uint8)
    target = np.ascontiguousarray(target, dtype=np.uint8)
    cap = query.shape[0] + target.shape[0] + 2
    cig = np.zeros(cap, np.uint32)
    n = C.c_int(0)
    sc = L.ref_ksw_global2(C.c_int(query.shape[0]), C.c_void_p(query.ctypes.data), C.c_int(target.shape[0]), C.c_void_p(target.ctypes.data), C.c_int(a),
                           C.c_int(b), C.c_int(o_del), C.c_int(e_del), C.c_int(o_ins), C.c_int(e_ins), C.c_int(int(w)), C.byref(n), C.c_void_p(cig.ctypes.data),
                           C.c_int(cap))
    assert n.value >= 0
    return sc, cig[:n.value].copy()


def gen_cigar2(fwd, query, rb, re, w_, a=1, b=4, o_del=6, e_del=1, o_ins=6, e_ins=1):
    """bwa_gen_cigar2 of the compiled reference on the genome `fwd` (codes 0..3; the shim packs it once per array):
    (score, cigar, NM, MD bytes) or None when the function returns no CIGAR."""
    L = stage_lib()
    L.ref_gen_cigar2.restype = C.c_int
    fwd = np.ascontiguousarray(fwd, dtype=np.uint8)
    query = np.ascontiguousarray(query, dtype=np.uint8)
    n = query.shape[0] + max(int(re - rb), 0)
    cig = np.zeros(n + 2, np.uint32)
    md = np.zeros(2 * n + 16, np.uint8)
    out = np.zeros(4, np.int32)
    rc = L.ref_gen_cigar2(C.c_void_p(fwd.ctypes.data), C.c_int64(fwd.shape[0]), C.c_int(a), C.c_int(b), C.c_int(o_del), C.c_int(e_del), C.c_int(o_ins), C.c_int(e_ins),
                          C.c_int(int(w_)), C.c_int(query.shape[0]), C.c_void_p(query.ctypes.data), C.c_int64(int(rb)), C.c_int64(int(re)), C.c_void_p(out.ctypes.data),
                          C.c_void_p(cig.ctypes.data), C.c_int(cig.shape[0]), C.c_void_p(md.ctypes.data), C.c_int(md.shape[0]))
    if rc == -1:
        return None
    assert rc == 0, rc
    return int(out[0]), cig[:int(out[1])].copy(), int(out[2]), md[:int(out[3])].tobytes()


def aln2sam(rec, blob, name, seq, qual, contig_blob, contig_off, softclip=0, rg_id=b""):
    """mem_aln2sam of the compiled reference for one record (oracle_py.SAM_REC_DTYPE scalar): the SAM text, bytes."""
    from oracle_py import SAM_REC_DTYPE
    L = stage_lib()
    L.ref_aln2sam.restype = C.c_int64
    rec = np.ascontiguousarray(rec, dtype=SAM_REC_DTYPE).reshape(1)
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    cap = len(name) + 2 * seq.shape[0] + blob.shape[0] * 4 + 512
    out = np.zeros(cap, np.uint8)
    q = C.c_char_p(qual) if qual is not None else C.c_char_p(None)
    n = L.ref_aln2sam(C.c_void_p(rec.ctypes.data), C.c_void_p(blob.ctypes.data), C.c_char_p(name), C.c_void_p(seq.ctypes.data), C.c_int(seq.shape[0]), q,
                      C.c_void_p(contig_blob.ctypes.data), C.c_void_p(contig_off.ctypes.data), C.c_int(contig_off.shape[0] - 1), C.c_int(int(softclip)), C.c_char_p(rg_id),
                      C.c_void_p(out.ctypes.data), C.c_int64(cap))
    assert n >= 0
    return out[:n].tobytes()


def kswv_batch(jobs, ref, qer, a=1, b=4, o_del=6, e_del=1, o_ins=6, e_ins=1):
    """sort_classify + mem_sam_pe_batch of the compiled reference (its AVX-512 kswv kernels) on the jobs: KSWR_DTYPE records."""
    from oracle_py import KSWV_JOB_DTYPE, KSWR_DTYPE
    L = stage_lib()
    L.ref_kswv_batch.restype = C.c_int
    jobs = np.ascontiguousarray(jobs, dtype=KSWV_JOB_DTYPE)
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    qer = np.ascontiguousarray(qer, dtype=np.uint8)
    out = np.zeros(jobs.shape[0], KSWR_DTYPE)
    rc = L.ref_kswv_batch(C.c_void_p(jobs.ctypes.data), C.c_int64(jobs.shape[0]), C.c_void_p(ref.ctypes.data), C.c_int64(ref.shape[0]), C.c_void_p(qer.ctypes.data),
                          C.c_int64(qer.shape[0]), C.c_int(a), C.c_int(b), C.c_int(o_del), C.c_int(e_del), C.c_int(o_ins), C.c_int(e_ins), C.c_void_p(out.ctypes.data))
    assert rc == 0
    return out


def matesw_pose(fwd, l_pac, contig_off, contig_len, reads, read_off, first, count, regs, reg_off, pes, a=1, pen_unpaired=17, max_matesw=50, min_seed_len=19):
    """The compiled reference's mem_sam_pe_batch_pre over one worker batch (oracle/ref_stage_shim.cpp ref_matesw_pose): (gar, jobs KSWV_JOB_DTYPE, ref bytes, query bytes)."""
    from oracle_py import KSWV_JOB_DTYPE, MATE_REG_DTYPE
    L = stage_lib()
    L.ref_matesw_pose.restype = C.c_int64
    fwd = np.ascontiguousarray(fwd, dtype=np.uint8)
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    read_off = np.ascontiguousarray(read_off, dtype=np.int64)
    regs = np.ascontiguousarray(regs, dtype=MATE_REG_DTYPE)
    reg_off = np.ascontiguousarray(reg_off, dtype=np.int64)
    contig_off = np.ascontiguousarray(contig_off, dtype=np.int64)
    contig_len = np.ascontiguousarray(contig_len, dtype=np.int32)
    pes = np.ascontiguousarray(np.asarray(pes, np.int32).reshape(4, 3))
    nrec = int(reg_off[first + count] - reg_off[first])
    cap = 4 * nrec + 64
    gar = np.zeros(cap, np.int32)
    jobs = np.zeros(cap, KSWV_JOB_DTYPE)
    ref = np.zeros(cap * 1400 + 4096, np.uint8)
    qer = np.zeros(cap * 520 + 4096, np.uint8)
    n_gar, rbytes, qbytes = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    n = L.ref_matesw_pose(C.c_void_p(fwd.ctypes.data), C.c_int64(l_pac), C.c_void_p(contig_off.ctypes.data), C.c_void_p(contig_len.ctypes.data), C.c_int(contig_off.shape[0]),
                          C.c_void_p(reads.ctypes.data), C.c_void_p(read_off.ctypes.data), C.c_int64(first), C.c_int64(count), C.c_void_p(regs.ctypes.data),
                          C.c_void_p(reg_off.ctypes.data), C.c_void_p(pes.ctypes.data), C.c_int(a), C.c_int(pen_unpaired), C.c_int(max_matesw), C.c_int(min_seed_len),
                          C.c_void_p(gar.ctypes.data), C.c_int64(cap), C.byref(n_gar), C.c_void_p(jobs.ctypes.data), C.c_int64(cap), C.c_void_p(ref.ctypes.data),
                          C.c_int64(ref.shape[0] - 4096), C.byref(rbytes), C.c_void_p(qer.ctypes.data), C.c_int64(qer.shape[0] - 4096), C.byref(qbytes))
    assert n >= 0
    return gar[:n_gar.value].copy(), jobs[:n].copy(), ref[:rbytes.value].copy(), qer[:qbytes.value].copy()
