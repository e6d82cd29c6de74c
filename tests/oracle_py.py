"""ctypes access to the C oracle (oracle/_build/libmeme_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class MemTl(C.Structure):
    _fields_ = [("start", C.c_int32), ("end", C.c_int32), ("hitbeg", C.c_int32),
                ("hitcount", C.c_int32), ("cache_refpos", C.c_uint64)]


MEM_TL_DTYPE = np.dtype([("start", "<i4"), ("end", "<i4"), ("hitbeg", "<i4"), ("hitcount", "<i4"),
                         ("cache_refpos", "<u8")])
SEQPAIR_DTYPE = np.dtype([(n, "<i4") for n in
                          ("idr", "idq", "id", "len1", "len2", "h0", "seqid", "regid", "score", "tle",
                           "gtle", "qle", "gscore", "max_off")])
assert MEM_TL_DTYPE.itemsize == 24 and SEQPAIR_DTYPE.itemsize == 56


class OrcIndex(C.Structure):
    _fields_ = [("text", C.c_void_p), ("sa", C.c_void_p), ("n", C.c_int64)]


class OrcSeedParams(C.Structure):
    _fields_ = [("min_seed_len", C.c_int32), ("split_len", C.c_int32), ("split_width", C.c_int32),
                ("max_mem_intv", C.c_int32), ("steps", C.c_int32)]


class OrcCounters(C.Structure):
    _fields_ = [("searches", C.c_int64), ("level_steps", C.c_int64), ("smems", C.c_int64),
                ("hits", C.c_int64)]


class OrcBswParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("o_del", "e_del", "o_ins", "e_ins", "zdrop", "end_bonus", "a", "b")]


def build():
    subprocess.run(["make", "-s", "-f", "oracle/Makefile"], cwd=REPO, check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(REPO, "oracle", "_build", "libmeme_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_seed_batch.restype = C.c_int
        _LIB.orc_bsw_batch.restype = None
        _LIB.orc_search.restype = C.c_uint32
    return _LIB


def default_seed_params(steps=3):
    return OrcSeedParams(19, 28, 10, 20, steps)


def default_bsw_params(end_bonus=5):
    # mem_opt_init (reference src/bwamem.cpp:126-162): a=1 b=4 o=6 e=1 zdrop=100 pen_clip=5
    return OrcBswParams(6, 1, 6, 1, 100, end_bonus, 1, 4)


class Index:
    """text = fwd+rc codes (uint8), sa = uint64 suffix array"""

    def __init__(self, text: np.ndarray, sa: np.ndarray):
        self.text = np.ascontiguousarray(text, dtype=np.uint8)
        self.sa = np.ascontiguousarray(sa, dtype=np.uint64)
        assert self.text.shape[0] == self.sa.shape[0]
        self.c = OrcIndex(self.text.ctypes.data, self.sa.ctypes.data, self.text.shape[0])


def seed_batch(index: Index, reads: np.ndarray, read_off: np.ndarray, params=None, smem_cap=512,
               hit_cap=1 << 16, threads=0):
    """Returns (smems [nreads, smem_cap] structured, n_smems, hits [nreads, hit_cap], n_hits, counters)."""
    params = params or default_seed_params()
    reads = np.ascontiguousarray(reads, dtype=np.uint8).reshape(-1)
    read_off = np.ascontiguousarray(read_off, dtype=np.int64)
    n = read_off.shape[0] - 1
    smems = np.zeros((n, smem_cap), dtype=MEM_TL_DTYPE)
    n_smems = np.zeros(n, dtype=np.int32)
    hits = np.zeros((n, hit_cap), dtype=np.uint64)
    n_hits = np.zeros(n, dtype=np.int64)
    ctr = OrcCounters()
    rc = lib().orc_seed_batch(C.byref(index.c), C.c_void_p(reads.ctypes.data), C.c_void_p(read_off.ctypes.data),
                              C.c_int64(n), C.byref(params), C.c_void_p(smems.ctypes.data), C.c_int32(smem_cap),
                              C.c_void_p(n_smems.ctypes.data), C.c_void_p(hits.ctypes.data), C.c_int64(hit_cap),
                              C.c_void_p(n_hits.ctypes.data), C.byref(ctr), C.c_int(threads))
    if rc != 0:
        raise RuntimeError("oracle capacity exceeded (raise smem_cap / hit_cap)")
    return smems, n_smems, hits, n_hits, ctr


REFPATH_COUNTERS = ("lookups", "partial_lookups", "compares", "extra_words", "hits", "smems", "searches")


def refpath_seed_batch(index: Index, l1, l2, reads, read_off, params=None, smem_cap=512, threads=0, keep_smems=True):
    """Instrumented restatement of the reference's probe sequence (oracle/meme_refpath.c).
    Returns (smems or None, n_smems, dict of work counters)."""
    params = params or default_seed_params()
    reads = np.ascontiguousarray(reads, dtype=np.uint8).reshape(-1)
    read_off = np.ascontiguousarray(read_off, dtype=np.int64)
    l1 = np.ascontiguousarray(l1)
    l2 = np.ascontiguousarray(l2)
    n = read_off.shape[0] - 1
    smems = np.zeros((n, smem_cap), dtype=MEM_TL_DTYPE) if keep_smems else None
    n_smems = np.zeros(n, dtype=np.int32)
    ctr = np.zeros(7, dtype=np.int64)
    rc = lib().rp_seed_batch(C.byref(index.c), C.c_void_p(l1.ctypes.data if l1.shape[0] else None),
                             C.c_void_p(l2.ctypes.data), C.c_int64(l2.shape[0]), C.c_void_p(reads.ctypes.data),
                             C.c_void_p(read_off.ctypes.data), C.c_int64(n), C.byref(params),
                             C.c_void_p(smems.ctypes.data if keep_smems else None), C.c_int32(smem_cap),
                             C.c_void_p(n_smems.ctypes.data), C.c_void_p(ctr.ctypes.data), C.c_int(threads))
    if rc != 0:
        raise RuntimeError("refpath capacity exceeded")
    return smems, n_smems, dict(zip(REFPATH_COUNTERS, ctr.tolist()))


def algorithmic_bytes(ctr, total_bases):
    """lookups*24 (+24 per partial hop) + compares*13 + extra words*8 + hits*(5+8) +
    SMEMs*24 + read bases (MODE 2/3 entry size 13 B; no ISA lookups in the restated MODE-2 path)."""
    return (24 * ctr["lookups"] + 24 * ctr["partial_lookups"] + 13 * ctr["compares"] + 8 * ctr["extra_words"] +
            13 * ctr["hits"] + 24 * ctr["smems"] + total_bases)


def tsc_hz():
    f = lib().orc_tsc_hz
    f.restype = C.c_double
    return f()


def load_prmi_files(prefix):
    dt = np.dtype([("icpt", "<f8"), ("slope", "<f8"), ("err", "<u8")])
    return (np.fromfile(prefix + ".suffixarray_uint64_L1_PARAMETERS", dtype=dt),
            np.fromfile(prefix + ".suffixarray_uint64_L2_PARAMETERS", dtype=dt))


def bsw_batch(pairs: np.ndarray, ref: np.ndarray, qer: np.ndarray, w: int, params=None, threads=0):
    """pairs: SEQPAIR_DTYPE array (modified in place). Returns number of DP cells evaluated."""
    params = params or default_bsw_params()
    assert pairs.dtype == SEQPAIR_DTYPE and pairs.flags.c_contiguous
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    qer = np.ascontiguousarray(qer, dtype=np.uint8)
    cells = C.c_int64(0)
    lib().orc_bsw_batch(C.c_void_p(pairs.ctypes.data), C.c_void_p(ref.ctypes.data), C.c_void_p(qer.ctypes.data),
                        C.c_int32(pairs.shape[0]), C.c_int32(w), C.byref(params), C.c_int(threads), C.byref(cells))
    return cells.value


def format_seed_dump(smems, n_smems, hits, first_id=0):
    """The `steps=4` text dump of the reference harness (test/Learned_seeding_big_read.cpp:286-299):
    per read "<id>:" then one "[start,end] [hit,hit,]" line per SMEM sorted by (start asc, end desc)."""
    out = []
    for r in range(n_smems.shape[0]):
        out.append("%d:" % (r + first_id))
        k = int(n_smems[r])
        s = smems[r, :k] if smems.ndim == 2 else smems[r][:k]
        order = sorted(range(k), key=lambda i: (int(s["start"][i]), -int(s["end"][i])))
        for i in order:
            hb, hc = int(s["hitbeg"][i]), int(s["hitcount"][i])
            hv = hits[r][hb:hb + hc]
            out.append("[%d,%d] [%s]" % (s["start"][i], s["end"][i], "".join("%d," % h for h in hv)))
    return "\n".join(out) + "\n"


def load_index_files(prefix: str) -> Index:
    text = np.fromfile(prefix + ".0123", dtype=np.uint8)
    raw = np.fromfile(prefix + ".pos_packed", dtype=np.uint8).reshape(-1, 5)
    sa = (raw[:, :4].copy().view("<u4").reshape(-1).astype(np.uint64) << np.uint64(8)) | raw[:, 4].astype(np.uint64)
    return Index(text, sa)


# ---- chaining (mem_chain_Learned + mem_chain_flt) --------------------------------------------------------------------------
class OrcChainOpt(C.Structure):
    _fields_ = [("w", C.c_int32), ("max_chain_gap", C.c_int32), ("max_occ", C.c_int32), ("min_seed_len", C.c_int32),
                ("min_chain_weight", C.c_int32), ("max_chain_extend", C.c_int32), ("mask_level", C.c_float), ("drop_ratio", C.c_float),
                ("l_pac", C.c_int64)]


ORC_CHAIN_DTYPE = np.dtype([("pos", "<i8"), ("rid", "<i4"), ("n_seeds", "<i4"), ("w", "<i4"), ("first", "<i4"), ("kept", "<i4"),
                            ("is_alt", "<i4"), ("seed_beg", "<i4"), ("_pad", "<i4")])
ORC_CSEED_DTYPE = np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4")])
assert ORC_CHAIN_DTYPE.itemsize == 40 and ORC_CSEED_DTYPE.itemsize == 16


def default_chain_opt(l_pac, w=100, max_chain_gap=10000, max_occ=500, min_seed_len=19, min_chain_weight=0, max_chain_extend=1 << 30,
                      mask_level=0.5, drop_ratio=0.5):
    # mem_opt_init, reference src/bwamem.cpp:126-162
    return OrcChainOpt(w, max_chain_gap, max_occ, min_seed_len, min_chain_weight, max_chain_extend, mask_level, drop_ratio, l_pac)


def chain_read(smems, hits, read_len, contig_off, contig_alt, opt, chain_cap=4096, seed_cap=1 << 16):
    """orc_chain_read for one read.  smems: MEM_TL_DTYPE array (hitbeg relative to `hits`); returns (rc, chains, seeds, tree_size,
    frac_rep) with rc = number of chains, -1 = undefined (equal B-tree keys), -2 = capacity."""
    L = lib()
    L.orc_chain_read.restype = C.c_int
    smems = np.ascontiguousarray(smems, dtype=MEM_TL_DTYPE)
    hits = np.ascontiguousarray(hits, dtype=np.uint64)
    contig_off = np.ascontiguousarray(contig_off, dtype=np.int64)
    contig_alt = np.ascontiguousarray(contig_alt, dtype=np.uint8)
    out = np.zeros(chain_cap, ORC_CHAIN_DTYPE)
    sd = np.zeros(seed_cap, ORC_CSEED_DTYPE)
    tree, frac = C.c_int(0), C.c_float(0)
    rc = L.orc_chain_read(C.c_void_p(smems.ctypes.data), C.c_int(smems.shape[0]), C.c_void_p(hits.ctypes.data), C.c_int(int(read_len)),
                          C.c_void_p(contig_off.ctypes.data), C.c_void_p(contig_alt.ctypes.data), C.c_int(contig_off.shape[0]), C.byref(opt),
                          C.c_void_p(out.ctypes.data), C.c_int(chain_cap), C.c_void_p(sd.ctypes.data), C.c_int(seed_cap), C.byref(tree), C.byref(frac))
    n = max(rc, 0)
    ns = int(out["n_seeds"][:n].sum())
    return rc, out[:n], sd[:ns], tree.value, frac.value


def chain_compare_batch(smems, smem_off, hits, hit_off, read_len, contig_off, contig_alt, opt, res, threads=0):
    """orc_chain_compare_batch: every read of a device chaining result (the dict hipapi's chain calls return) against orc_chain_read.
    Returns (number of differing reads, first differing read or -1)."""
    L = lib()
    L.orc_chain_compare_batch.restype = C.c_int64
    smems = np.ascontiguousarray(smems, dtype=MEM_TL_DTYPE)
    smem_off = np.ascontiguousarray(smem_off, dtype=np.int64)
    hits = np.ascontiguousarray(hits, dtype=np.uint64)
    hit_off = np.ascontiguousarray(hit_off, dtype=np.int64)
    read_len = np.ascontiguousarray(read_len, dtype=np.int32)
    contig_off = np.ascontiguousarray(contig_off, dtype=np.int64)
    contig_alt = np.ascontiguousarray(contig_alt, dtype=np.uint8)
    ch = np.ascontiguousarray(res["chains"]); sd = np.ascontiguousarray(res["seeds"])
    assert ch.dtype.itemsize == 40 and sd.dtype.itemsize == 16
    co = np.ascontiguousarray(res["chain_off"], dtype=np.int64); so = np.ascontiguousarray(res["seed_off"], dtype=np.int64)
    tree = np.ascontiguousarray(res["tree_size"], dtype=np.int32); frac = np.ascontiguousarray(res["frac_rep"], dtype=np.float32)
    first = C.c_int64(-1)
    p = lambda a: C.c_void_p(a.ctypes.data)
    nbad = L.orc_chain_compare_batch(p(smems), p(smem_off), p(hits), p(hit_off), p(read_len), C.c_int64(read_len.shape[0]), p(contig_off), p(contig_alt),
                                     C.c_int(contig_off.shape[0]), C.byref(opt), p(co), p(ch), p(so), p(sd), p(tree), p(frac), C.c_int(threads), C.byref(first))
    return int(nbad), int(first.value)


# ---- seed extension stage ---------------------------------------------------------------------------------------------------------
class OrcExtOpt(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "pen_clip5", "pen_clip3", "w", "zdrop")]


ORC_ALNREG_DTYPE = np.dtype([("rb", "<i8"), ("re", "<i8")] + [(n, "<i4") for n in ("qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w",
                                                                                   "seedcov", "secondary", "secondary_all", "seedlen0", "n_comp", "is_alt")] +
                            [("frac_rep", "<f4"), ("pad", "<i4")])
assert ORC_ALNREG_DTYPE.itemsize == 88
ALNREG_FIELDS = ("rb", "re", "qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w", "seedcov", "secondary", "secondary_all", "seedlen0")


def default_ext_opt(w=100):
    return OrcExtOpt(1, 4, 6, 1, 6, 1, 5, 5, w, 100)


def chains_as_orc(chains):
    """meme_chain / fixture rows -> ORC_CHAIN_DTYPE records"""
    out = np.zeros(chains.shape[0], ORC_CHAIN_DTYPE)
    for f in ("pos", "rid", "n_seeds", "w", "first", "kept", "is_alt", "seed_beg"):
        out[f] = chains[f]
    return out


def extend_batch(reads, read_off, chain_off, chains, seed_off, seeds, frac_rep, text, l_pac, contig_off, contig_len, opt=None, threads=0, seed_score=None):
    """orc_extend_batch[_scored]: records of every read's chained seeds (ORC_ALNREG_DTYPE, indexed by seed_off) + (jobs, retried).
    seed_score: the seeds' scores as the seed filter leaves them (None: score = length)."""
    L = lib()
    L.orc_extend_batch_scored.restype = C.c_int
    opt = opt or default_ext_opt()
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    read_off = np.ascontiguousarray(read_off, dtype=np.int64)
    chain_off = np.ascontiguousarray(chain_off, dtype=np.int64)
    chains = np.ascontiguousarray(chains, dtype=ORC_CHAIN_DTYPE)
    seed_off = np.ascontiguousarray(seed_off, dtype=np.int64)
    seeds = np.ascontiguousarray(seeds, dtype=ORC_CSEED_DTYPE)
    frac_rep = np.ascontiguousarray(frac_rep, dtype=np.float32)
    text = np.ascontiguousarray(text, dtype=np.uint8)
    contig_off = np.ascontiguousarray(contig_off, dtype=np.int64)
    contig_len = np.ascontiguousarray(contig_len, dtype=np.int32)
    out = np.zeros(int(seed_off[-1]), ORC_ALNREG_DTYPE)
    stats = np.zeros(2, np.int64)
    p = lambda a: C.c_void_p(a.ctypes.data)
    if seed_score is not None:
        seed_score = np.ascontiguousarray(seed_score, dtype=np.int32)
    rc = L.orc_extend_batch_scored(p(reads), p(read_off), C.c_int64(read_off.shape[0] - 1), p(chain_off), p(chains), p(seed_off), p(seeds),
                            p(seed_score) if seed_score is not None else C.c_void_p(0), p(frac_rep), p(text),
                            C.c_int64(int(l_pac)), p(contig_off), p(contig_len), C.byref(opt), p(out), C.c_int(threads), p(stats))
    assert rc == 0
    return out, (int(stats[0]), int(stats[1]))


def _repack_filtered(chains, seed_off, seeds, score, kept):
    """What the in-place filters leave (a read's surviving seeds at the front of its old range) -> (chains, seed_off, seeds, score) packed."""
    new_off = np.zeros(seed_off.shape[0], np.int64)
    new_off[1:] = np.cumsum(kept)
    idx = np.repeat(seed_off[:-1] - new_off[:-1], kept) + np.arange(int(new_off[-1]))
    return chains, new_off, seeds[idx].copy(), score[idx].copy()


def flt_batch(reads, read_off, chain_off, chains, seed_off, seeds, text, l_pac, contig_off, contig_len, opt=None, min_chain_weight=0, threads=0):
    """orc_flt_batch (mem_flt_chained_seeds): (chains, seed_off, seeds, score) after the filter + the number of alignments it ran."""
    L = lib()
    L.orc_flt_batch.restype = C.c_int
    opt = opt or default_ext_opt()
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    read_off = np.ascontiguousarray(read_off, dtype=np.int64)
    chain_off = np.ascontiguousarray(chain_off, dtype=np.int64)
    chains = np.array(chains, dtype=ORC_CHAIN_DTYPE)
    seed_off = np.ascontiguousarray(seed_off, dtype=np.int64)
    seeds = np.array(seeds, dtype=ORC_CSEED_DTYPE)
    text = np.ascontiguousarray(text, dtype=np.uint8)
    contig_off = np.ascontiguousarray(contig_off, dtype=np.int64)
    contig_len = np.ascontiguousarray(contig_len, dtype=np.int32)
    n = read_off.shape[0] - 1
    score = np.zeros(seeds.shape[0], np.int32)
    kept = np.zeros(n, np.int64)
    n_sw = np.zeros(1, np.int64)
    p = lambda a: C.c_void_p(a.ctypes.data)
    rc = L.orc_flt_batch(p(reads), p(read_off), C.c_int64(n), p(chain_off), p(chains), p(seed_off), p(seeds), p(score), p(text), C.c_int64(int(l_pac)), p(contig_off),
                         p(contig_len), C.c_int(contig_off.shape[0]), C.byref(opt), C.c_int(int(min_chain_weight)), p(kept), C.c_int(threads), p(n_sw))
    assert rc == 0
    return _repack_filtered(chains, seed_off, seeds, score, kept) + (int(n_sw[0]),)


def ksw_global2(query, target, w, a=1, b=4, o_del=6, e_del=1, o_ins=6, e_ins=1):
    """orc_ksw_global2: (score, cigar as uint32 array) of the banded global alignment of two code arrays."""
    L = lib()
    L.orc_ksw_global2.restype = C.c_int
    query = np.ascontiguousarray(query, dtype=np.uint8)
    target = np.ascontiguousarray(target, dtype=np.uint8)
    cig = np.zeros(query.shape[0] + target.shape[0] + 2, np.uint32)
    n = C.c_int(0)
    sc = L.orc_ksw_global2(C.c_int(query.shape[0]), C.c_void_p(query.ctypes.data), C.c_int(target.shape[0]), C.c_void_p(target.ctypes.data), C.c_int(a),
                           C.c_int(b), C.c_int(o_del), C.c_int(e_del), C.c_int(o_ins), C.c_int(e_ins), C.c_int(int(w)), C.byref(n), C.c_void_p(cig.ctypes.data))
    return sc, cig[:n.value].copy()


def gen_cigar2(text, l_pac, query, rb, re, w_, a=1, b=4, o_del=6, e_del=1, o_ins=6, e_ins=1):
    """orc_gen_cigar2 (bwa_gen_cigar2 whole): (score, cigar uint32 array, NM, MD bytes) or None for a call the function rejects."""
    L = lib()
    L.orc_gen_cigar2.restype = C.c_int
    text = np.ascontiguousarray(text, dtype=np.uint8)
    query = np.ascontiguousarray(query, dtype=np.uint8)
    n = query.shape[0] + max(int(re - rb), 0)
    cig = np.zeros(n + 2, np.uint32)
    md = np.zeros(2 * n + 16, np.uint8)
    sc, nc, nm = C.c_int(0), C.c_int(0), C.c_int(0)
    rc = L.orc_gen_cigar2(C.c_void_p(text.ctypes.data), C.c_int64(int(l_pac)), C.c_int(a), C.c_int(b), C.c_int(o_del), C.c_int(e_del), C.c_int(o_ins), C.c_int(e_ins),
                          C.c_int(int(w_)), C.c_int(query.shape[0]), C.c_void_p(query.ctypes.data), C.c_int64(int(rb)), C.c_int64(int(re)), C.byref(sc), C.byref(nc),
                          C.c_void_p(cig.ctypes.data), C.byref(nm), C.c_void_p(md.ctypes.data))
    if rc != 0:
        return None
    return sc.value, cig[:nc.value].copy(), nm.value, md[:int(np.argmin(md != 0))].tobytes()


SAM_REC_DTYPE = np.dtype([("pos", "<i8"), ("m_pos", "<i8"), ("cigar_off", "<i8"), ("m_cigar_off", "<i8"), ("xa_off", "<i8")] +
                         [(n, "<i4") for n in ("read", "flag", "rid", "is_rev", "is_alt", "mapq", "NM", "score", "sub", "n_cigar", "has_mate", "m_rid", "m_is_rev", "m_is_alt",
                                               "m_n_cigar", "which")])
assert SAM_REC_DTYPE.itemsize == 104


def contig_table(names):
    blob = b"".join(n.encode() if isinstance(n, str) else n for n in names)
    off = np.zeros(len(names) + 1, np.int32)
    off[1:] = np.cumsum([len(n) for n in names])
    return np.frombuffer(blob, dtype=np.uint8).copy(), off


def aln2sam(rec, blob, name, seq, qual, contig_blob, contig_off, softclip=0, rg_id=b""):
    """orc_aln2sam: the SAM text of one record (SAM_REC_DTYPE scalar), bytes."""
    L = lib()
    L.orc_aln2sam.restype = C.c_int64
    rec = np.ascontiguousarray(rec, dtype=SAM_REC_DTYPE).reshape(1)
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    out = np.zeros(len(name) + 2 * seq.shape[0] + blob.shape[0] * 4 + 512, np.uint8)
    q = C.c_char_p(qual) if qual is not None else C.c_char_p(None)
    n = L.orc_aln2sam(C.c_void_p(rec.ctypes.data), C.c_void_p(blob.ctypes.data), C.c_char_p(name), C.c_int(len(name)), C.c_void_p(seq.ctypes.data), C.c_int(seq.shape[0]), q,
                      C.c_void_p(contig_blob.ctypes.data), C.c_void_p(contig_off.ctypes.data), C.c_int(int(softclip)), C.c_char_p(rg_id), C.c_void_p(out.ctypes.data))
    return out[:n].tobytes()


KSWV_JOB_DTYPE = np.dtype([("idr", "<i8"), ("idq", "<i8"), ("len1", "<i4"), ("len2", "<i4"), ("xtra", "<i4"), ("pad", "<i4")])
KSWR_DTYPE = np.dtype([(n, "<i4") for n in ("score", "te", "qe", "score2", "te2", "tb", "qb")])
KSW_XBYTE, KSW_XSTOP, KSW_XSUBO, KSW_XSTART = 0x10000, 0x20000, 0x40000, 0x80000


def kswv_batch(jobs, ref, qer, a=1, b=4, o_del=6, e_del=1, o_ins=6, e_ins=1, threads=0):
    """orc_kswv_batch: kswr_t records (KSWR_DTYPE) of the mate-rescue Smith-Waterman jobs (KSWV_JOB_DTYPE), and the DP cells evaluated."""
    L = lib()
    jobs = np.ascontiguousarray(jobs, dtype=KSWV_JOB_DTYPE)
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    qer = np.ascontiguousarray(qer, dtype=np.uint8)
    out = np.zeros(jobs.shape[0], KSWR_DTYPE)
    cells = C.c_int64(0)
    L.orc_kswv_batch(C.c_void_p(jobs.ctypes.data), C.c_int64(jobs.shape[0]), C.c_void_p(ref.ctypes.data), C.c_void_p(qer.ctypes.data), C.c_int(a), C.c_int(b),
                     C.c_int(o_del), C.c_int(e_del), C.c_int(o_ins), C.c_int(e_ins), C.c_void_p(out.ctypes.data), C.c_int(threads or (os.cpu_count() or 1)),
                     C.byref(cells))
    return out, cells.value


# ---- mate rescue, the posing step (orc_matesw_pose: mem_sam_pe_batch_pre + mem_matesw_batch_pre, reference src/bwamem_pair.cpp:660-716, 1060-1223) ----
MATE_REG_DTYPE = np.dtype([("rb", "<i8"), ("rid", "<i4"), ("score", "<i4")])
MATE_JOB_DTYPE = np.dtype([("rb", "<i8"), ("read", "<i4"), ("len1", "<i4"), ("len2", "<i4"), ("xtra", "<i4"), ("is_rev", "<i4"), ("pad", "<i4")])
assert MATE_REG_DTYPE.itemsize == 16 and MATE_JOB_DTYPE.itemsize == 32


def matesw_pose(regs, reg_off, first, count, read_len, pes, l_pac, contig_off, contig_len, a=1, pen_unpaired=17, max_matesw=50, min_seed_len=19):
    """One worker batch: reads [first, first + count).  pes: 4 x (low, high, failed).  Returns (gar int32 array, jobs MATE_JOB_DTYPE)."""
    L = lib()
    L.orc_matesw_pose.restype = C.c_int64
    regs = np.ascontiguousarray(regs, dtype=MATE_REG_DTYPE)
    reg_off = np.ascontiguousarray(reg_off, dtype=np.int64)
    read_len = np.ascontiguousarray(read_len, dtype=np.int32)
    pes4 = np.zeros((4, 4), np.int32)
    pes4[:, :3] = np.asarray(pes, np.int32).reshape(4, 3)
    contig_off = np.ascontiguousarray(contig_off, dtype=np.int64)
    contig_len = np.ascontiguousarray(contig_len, dtype=np.int32)
    opt = np.array([a, pen_unpaired, max_matesw, min_seed_len], np.int32)
    nrec = int(reg_off[first + count] - reg_off[first])
    gar = np.full(4 * nrec + 8, -7, np.int32)
    jobs = np.zeros(4 * nrec + 8, MATE_JOB_DTYPE)
    n_gar = C.c_int64(0)
    n = L.orc_matesw_pose(C.c_void_p(regs.ctypes.data), C.c_void_p(reg_off.ctypes.data), C.c_int64(first), C.c_int64(count), C.c_void_p(read_len.ctypes.data),
                          C.c_void_p(pes4.ctypes.data), C.c_int64(l_pac), C.c_void_p(contig_off.ctypes.data), C.c_void_p(contig_len.ctypes.data), C.c_int(contig_off.shape[0]),
                          C.c_void_p(opt.ctypes.data), C.c_void_p(gar.ctypes.data), C.c_int64(gar.shape[0]), C.byref(n_gar), C.c_void_p(jobs.ctypes.data), C.c_int64(jobs.shape[0]))
    assert n >= 0
    return gar[:n_gar.value].copy(), jobs[:n].copy()


def matesw_job_seqs(jobs, text, reads, read_off):
    """The jobs' sequences back to back, as the step stores them: (KSWV_JOB_DTYPE jobs with idr / idq, ref bytes, query bytes)."""
    L = lib()
    L.orc_matesw_job_seqs.restype = None
    jobs = np.ascontiguousarray(jobs, dtype=MATE_JOB_DTYPE)
    text = np.ascontiguousarray(text, dtype=np.uint8)
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    read_off = np.ascontiguousarray(read_off, dtype=np.int64)
    kj = np.zeros(jobs.shape[0], KSWV_JOB_DTYPE)
    kj["len1"], kj["len2"], kj["xtra"] = jobs["len1"], jobs["len2"], jobs["xtra"]
    kj["idr"][1:] = np.cumsum(jobs["len1"].astype(np.int64))[:-1]
    kj["idq"][1:] = np.cumsum(jobs["len2"].astype(np.int64))[:-1]
    ref = np.zeros(int(jobs["len1"].astype(np.int64).sum()) + 8, np.uint8)
    qer = np.zeros(int(jobs["len2"].astype(np.int64).sum()) + 8, np.uint8)
    for k in range(jobs.shape[0]):
        L.orc_matesw_job_seqs(C.c_void_p(jobs[k:k + 1].ctypes.data), C.c_void_p(text.ctypes.data), C.c_void_p(reads.ctypes.data), C.c_void_p(read_off.ctypes.data),
                              C.c_void_p(ref.ctypes.data + int(kj["idr"][k])), C.c_void_p(qer.ctypes.data + int(kj["idq"][k])))
    return kj, ref[:-8], qer[:-8]
