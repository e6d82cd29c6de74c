"""ctypes bindings of the C ABI in include/meme_hip.h (libmeme_hip.so, built in-tree by
`make -C bwa-meme_amd hip`).  Fails loudly when the library or a HIP device is missing -- there is no
CPU fallback on the product path."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("MEME_HIP_LIB") or os.path.join(PKG, "libmeme_hip.so")

MEM_TL = np.dtype([("start", "<i4"), ("end", "<i4"), ("hitbeg", "<i4"), ("hitcount", "<i4"),
                   ("cache_refpos", "<u8")])
SEQPAIR = np.dtype([(n, "<i4") for n in ("idr", "idq", "id", "len1", "len2", "h0", "seqid", "regid", "score",
                                         "tle", "gtle", "qle", "gscore", "max_off")])


class SeedOpt(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("min_seed_len", "split_len", "split_width", "max_mem_intv", "rounds",
                                         "hits_per_smem")]


class BswOpt(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("o_del", "e_del", "o_ins", "e_ins", "zdrop", "end_bonus", "a", "b")]


class IndexArrays(C.Structure):
    _fields_ = [("sa_num", C.c_int64), ("d_sa_ent", C.c_void_p), ("d_pac64", C.c_void_p), ("d_l2", C.c_void_p),
                ("l2_records", C.c_int64), ("d_l1", C.c_void_p), ("l1_records", C.c_int64)]


class SeedHostResult(C.Structure):
    _fields_ = [("smems", C.c_void_p), ("smem_off", C.c_void_p), ("hits", C.c_void_p), ("hit_off", C.c_void_p),
                ("total_smems", C.c_int64), ("total_hits", C.c_int64)]


class ChainOpt(C.Structure):
    _fields_ = [("w", C.c_int32), ("max_chain_gap", C.c_int32), ("max_occ", C.c_int32), ("min_seed_len", C.c_int32),
                ("min_chain_weight", C.c_int32), ("max_chain_extend", C.c_int32), ("mask_level", C.c_float), ("drop_ratio", C.c_float),
                ("l_pac", C.c_int64)]


class Contig(C.Structure):
    _fields_ = [("offset", C.c_int64), ("len", C.c_int32), ("is_alt", C.c_int32)]


class ChainHostResult(C.Structure):
    _fields_ = [("nreads", C.c_int64), ("chain_off", C.c_void_p), ("chains", C.c_void_p), ("seed_off", C.c_void_p), ("seeds", C.c_void_p),
                ("tree_size", C.c_void_p), ("frac_rep", C.c_void_p), ("fallback", C.c_void_p), ("total_chains", C.c_int64),
                ("total_seeds", C.c_int64), ("n_fallback", C.c_int64), ("n_tier2", C.c_int64)]


class ExtOpt(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "pen_clip5", "pen_clip3", "w", "zdrop")]


class ExtHostResult(C.Structure):
    _fields_ = [("nreads", C.c_int64), ("reg_off", C.c_void_p), ("regs", C.c_void_p), ("total_regs", C.c_int64), ("total_chains", C.c_int64),
                ("n_pairs", C.c_int64), ("n_retried", C.c_int64), ("n_bsw_calls", C.c_int64), ("n_tier2", C.c_int64), ("chain_ms", C.c_float),
                ("ext_ms", C.c_float), ("bsw_ms", C.c_float), ("n_flt_jobs", C.c_int64), ("n_flt_dropped", C.c_int64), ("n_exact_prefix", C.c_int64),
                ("census_band_cells", C.c_int64), ("census_class", C.c_int64 * 9), ("total_seeds", C.c_int64), ("n_ext_seeds", C.c_int64)]


def default_ext_opt(w=100):
    # mem_opt_init (reference src/bwamem.cpp:126-162): a 1, b 4, o_del = o_ins 6, e_del = e_ins 1, pen_clip5 = pen_clip3 5, w 100, zdrop 100
    return ExtOpt(1, 4, 6, 1, 6, 1, 5, 5, w, 100)


# = meme_alnreg = mem_alnreg_t (112 bytes)
ALNREG = np.dtype({"names": ["rb", "re", "qb", "qe", "rid", "c", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w", "seedcov", "secondary",
                             "secondary_all", "seedlen0", "n_comp_is_alt", "frac_rep", "hash", "flg"],
                   "formats": ["<i8", "<i8", "<i4", "<i4", "<i4", "<u8"] + ["<i4"] * 12 + ["<f4", "<u8", "<i4"],
                   "offsets": [0, 8, 16, 20, 24, 32] + [40 + 4 * k for k in range(12)] + [88, 96, 104], "itemsize": 112})

def records_equal(a, b):
    """Field-by-field equality of two arrays of one structured dtype (numpy does not copy the padding bytes of a padded record type, so the
    raw bytes of two equal arrays differ in whatever the allocator left there)."""
    return a.dtype == b.dtype and a.shape == b.shape and all(np.array_equal(a[f], b[f]) for f in a.dtype.names)


GJOB = np.dtype([("rb", "<i8"), ("read", "<i4"), ("qb", "<i4"), ("qlen", "<i4"), ("tlen", "<i4"), ("w", "<i4"), ("rev", "<i4")])
GRES = np.dtype([("score", "<i4"), ("n_cigar", "<i4"), ("cigar_off", "<i8")])
assert GJOB.itemsize == 32 and GRES.itemsize == 16


class GresHost(C.Structure):
    _fields_ = [("njobs", C.c_int64), ("res", C.c_void_p), ("cigars", C.c_void_p), ("total_ops", C.c_int64), ("kernel_ms", C.c_float)]


CJOB = np.dtype([("rb", "<i8"), ("read", "<i4"), ("qb", "<i4"), ("qlen", "<i4"), ("tlen", "<i4"), ("w_", "<i4"), ("pad", "<i4")])
CRES = np.dtype([("score", "<i4"), ("n_cigar", "<i4"), ("nm", "<i4"), ("md_len", "<i4"), ("cigar_off", "<i8"), ("md_off", "<i8")])
assert CJOB.itemsize == 32 and CRES.itemsize == 32


class CresHost(C.Structure):
    _fields_ = [("njobs", C.c_int64), ("res", C.c_void_p), ("cigars", C.c_void_p), ("total_ops", C.c_int64), ("md", C.c_void_p), ("md_bytes", C.c_int64),
                ("kernel_ms", C.c_float)]


SAM_REC = np.dtype([("pos", "<i8"), ("m_pos", "<i8"), ("cigar_off", "<i8"), ("m_cigar_off", "<i8"), ("xa_off", "<i8")] +
                   [(n, "<i4") for n in ("read", "flag", "rid", "is_rev", "is_alt", "mapq", "NM", "score", "sub", "n_cigar", "has_mate", "m_rid", "m_is_rev", "m_is_alt",
                                         "m_n_cigar", "which")])
assert SAM_REC.itemsize == 104


class SamHost(C.Structure):
    _fields_ = [("nrecs", C.c_int64), ("text_off", C.c_void_p), ("text", C.c_void_p), ("text_bytes", C.c_int64), ("kernel_ms", C.c_float)]


class KswvHost(C.Structure):
    _fields_ = [("njobs", C.c_int64), ("res", C.c_void_p), ("kernel_ms", C.c_float)]


class MateHost(C.Structure):
    _fields_ = [("nreads", C.c_int64), ("nbatches", C.c_int64), ("njobs", C.c_int64), ("n_gar", C.c_int64), ("gar", C.c_void_p), ("gar_off", C.c_void_p), ("job_off", C.c_void_p),
                ("jobs", C.c_void_p), ("res", C.c_void_p), ("pose_ms", C.c_float), ("kernel_ms", C.c_float)]


class MateOpt(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "pen_unpaired", "max_matesw", "min_seed_len", "batch_reads")]


MATE_REG = np.dtype([("rb", "<i8"), ("rid", "<i4"), ("score", "<i4")])
KSWV_JOB = np.dtype([("idr", "<i8"), ("idq", "<i8"), ("len1", "<i4"), ("len2", "<i4"), ("xtra", "<i4"), ("pad", "<i4")])
KSWR = np.dtype([(n, "<i4") for n in ("score", "te", "qe", "score2", "te2", "tb", "qb")])
CHAIN = np.dtype({"names": ["pos", "rid", "n_seeds", "w", "first", "kept", "is_alt", "seed_beg"],
                  "formats": ["<i8", "<i4", "<i4", "<i4", "<i4", "<i2", "<i2", "<i4"], "offsets": [0, 8, 12, 16, 20, 24, 26, 28], "itemsize": 40})
CHAIN_SEED = np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4")])


def default_chain_opt(l_pac):
    # mem_opt_init (reference src/bwamem.cpp:126-162): w 100, max_chain_gap 10000, max_occ 500, min_seed_len 19,
    # min_chain_weight 0, max_chain_extend 1<<30, mask_level 0.5, drop_ratio 0.5
    return ChainOpt(100, 10000, 500, 19, 0, 1 << 30, 0.5, 0.5, l_pac)


class SeedResult(C.Structure):
    _fields_ = [("d_smems", C.c_void_p), ("d_smem_off", C.c_void_p), ("d_hits", C.c_void_p),
                ("d_hit_off", C.c_void_p), ("total_smems", C.c_int64), ("total_hits", C.c_int64),
                ("searches", C.c_int64)]


class Timings(C.Structure):
    _fields_ = [("seed_kernel_ms", C.c_float), ("seed_gather_ms", C.c_float), ("bsw_kernel_ms", C.c_float),
                ("seed_launches", C.c_int64), ("bsw_launches", C.c_int64), ("seed_pack_ms", C.c_float),
                ("seed_windows", C.c_int64), ("chain_kernel_ms", C.c_float), ("chain_pass2_ms", C.c_float), ("chain_tier3_ms", C.c_float),
                ("chain_tier2_reads", C.c_int64), ("chain_tier3_reads", C.c_int64),
                ("seed_reseed_ms", C.c_float), ("seed_lane_searches", C.c_int64), ("gcig_class_jobs", C.c_int64 * 6)]


# every symbol include/meme_hip.h declares (tests/test_abi.py checks the library exports them all)
EXPORTS = ["meme_device_count", "meme_ctx_create", "meme_ctx_destroy", "meme_last_error", "meme_ctx_sync",
           "meme_ctx_stream", "meme_index_load_host", "meme_index_load_files", "meme_index_pac64_words",
           "meme_index_pos5_bytes",
           "meme_index_attach", "meme_index_describe", "meme_index_share", "meme_index_replicate", "meme_host_alloc",
           "meme_host_free", "meme_stage_pack_text", "meme_stage_pos5_from_sa", "meme_stage_build_entries", "meme_stage_build_plcp",
           "meme_stage_entries_from_sa", "meme_stage_rmi32", "meme_sa_build_device", "meme_prmi_train_device", "meme_seed_batch", "meme_seed_batch_host", "meme_seed_batch_resident", "meme_seed_batch_resident_ascii", "meme_seed_reserve", "meme_chain_last_batch_host", "meme_chain_batch_host", "meme_extend_last_batch_host", "meme_global_batch_host", "meme_gen_cigar_batch_host", "meme_sam_stage_text", "meme_sam_format_batch_host", "meme_kswv_batch_host", "meme_matesw_batch_host", "meme_seed_batch_device",
           "meme_bsw_batch", "meme_bsw_batch_device", "meme_get_timings", "meme_set_tuning"]

_lib = None


def default_seed_opt(rounds=3, hits_per_smem=0):
    # mem_opt_init (reference src/bwamem.cpp:126-162): min_seed_len 19, split_factor 1.5, split_width 10,
    # max_mem_intv 20
    return SeedOpt(19, 28, 10, 20, rounds, hits_per_smem)


def default_bsw_opt(end_bonus=5):
    return BswOpt(6, 1, 6, 1, 100, end_bonus, 1, 4)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: build it with `make -C bwa-meme_amd hip` "
                               "(or __graft_entry__.build())" % LIB_PATH)
        try:
            # torch ships its own copy of the HIP runtime: when both live in one process, torch's has to be loaded first
            # (with ours first, torch finds no GPU)
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.meme_ctx_create.restype = C.c_void_p
        L.meme_ctx_create.argtypes = [C.c_int]
        L.meme_ctx_destroy.argtypes = [C.c_void_p]
        L.meme_ctx_destroy.restype = None
        L.meme_last_error.restype = C.c_char_p
        L.meme_ctx_stream.restype = C.c_void_p
        L.meme_ctx_stream.argtypes = [C.c_void_p]
        for f in ("meme_index_pac64_words", "meme_index_pos5_bytes"):
            getattr(L, f).restype = C.c_int64
            getattr(L, f).argtypes = [C.c_int64]
        L.meme_host_alloc.restype = C.c_void_p
        L.meme_host_alloc.argtypes = [C.c_int64]
        L.meme_host_free.argtypes = [C.c_void_p]
        L.meme_host_free.restype = None
        _lib = L
    return _lib


class MemeError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise MemeError("meme_hip error %d: %s" % (rc, lib().meme_last_error().decode()))


def _p(a):
    return C.c_void_p(a.ctypes.data)


class Context:
    """One HIP device + stream + workspaces (meme_ctx)."""

    def __init__(self, device=0):
        L = lib()
        if L.meme_device_count() <= 0:
            raise MemeError("no HIP device visible -- the MI355X backend has no CPU fallback")
        self.h = L.meme_ctx_create(device)
        if not self.h:
            raise MemeError(L.meme_last_error().decode())
        self.device = device
        # MEME_TUNING="key=value,key=value": tuning switches for every ctx of the process (probes and A/B runs of the tests and bench.py)
        for kv in filter(None, os.environ.get("MEME_TUNING", "").split(",")):
            k, v = kv.split("=")
            self.set_tuning(k.strip(), int(v))

    def close(self):
        if getattr(self, "h", None):
            lib().meme_ctx_destroy(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- index ---------------------------------------------------------------------------------------
    def load_index_files(self, prefix):
        _check(lib().meme_index_load_files(C.c_void_p(self.h), prefix.encode()))

    def load_index_host(self, pos_packed, text, l1, l2):
        pos_packed = np.ascontiguousarray(pos_packed, dtype=np.uint8)
        text = np.ascontiguousarray(text, dtype=np.uint8)
        l1 = np.ascontiguousarray(l1).view(np.uint8)
        l2 = np.ascontiguousarray(l2).view(np.uint8)
        _check(lib().meme_index_load_host(C.c_void_p(self.h), _p(pos_packed), C.c_int64(text.shape[0]), _p(text),
                                          _p(l1), C.c_int64(l1.shape[0]), _p(l2), C.c_int64(l2.shape[0])))

    def attach_index(self, arrays: IndexArrays):
        _check(lib().meme_index_attach(C.c_void_p(self.h), C.byref(arrays)))

    def describe_index(self) -> IndexArrays:
        a = IndexArrays()
        _check(lib().meme_index_describe(C.c_void_p(self.h), C.byref(a)))
        return a

    def replicate_index_from(self, src: "Context"):
        """meme_index_replicate: device-to-device copy of src's staged index (same device: shared)."""
        _check(lib().meme_index_replicate(C.c_void_p(self.h), C.c_void_p(src.h)))

    def set_tuning(self, key, value):
        _check(lib().meme_set_tuning(C.c_void_p(self.h), key.encode(), C.c_int64(value)))

    def timings(self) -> Timings:
        t = Timings()
        _check(lib().meme_get_timings(C.c_void_p(self.h), C.byref(t)))
        return t

    def sync(self):
        _check(lib().meme_ctx_sync(C.c_void_p(self.h)))

    # ---- seeding (host buffers in, host buffers out) ----------------------------------------------------
    def seed_batch(self, reads, read_off, opt=None, smem_capacity=None, hit_capacity=None):
        opt = opt or default_seed_opt()
        reads = np.ascontiguousarray(reads, dtype=np.uint8).reshape(-1)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        n = read_off.shape[0] - 1
        smem_capacity = smem_capacity or max(64 * n, 1024)
        hit_capacity = hit_capacity or max(1024 * n, 1 << 16)
        while True:
            smems = np.zeros(smem_capacity, dtype=MEM_TL)
            hits = np.zeros(hit_capacity, dtype=np.uint64)
            smem_off = np.zeros(n + 1, dtype=np.int64)
            hit_off = np.zeros(n + 1, dtype=np.int64)
            ts, th = C.c_int64(0), C.c_int64(0)
            rc = lib().meme_seed_batch(C.c_void_p(self.h), _p(reads), _p(read_off), C.c_int64(n), C.byref(opt),
                                       _p(smems), C.c_int64(smem_capacity), _p(smem_off), _p(hits),
                                       C.c_int64(hit_capacity), _p(hit_off), C.byref(ts), C.byref(th))
            if rc == -4:  # MEME_E_CAPACITY: sizes were reported back
                smem_capacity = max(smem_capacity, ts.value + 1)
                hit_capacity = max(hit_capacity, th.value + 1)
                continue
            _check(rc)
            return smems[:ts.value], smem_off, hits[:th.value], hit_off

    def seed_batch_host(self, reads, read_off, opt=None):
        """meme_seed_batch_host: results come back in pinned buffers owned by the ctx (copied out here)."""
        opt = opt or default_seed_opt()
        reads = np.ascontiguousarray(reads, dtype=np.uint8).reshape(-1)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        n = read_off.shape[0] - 1
        res = SeedHostResult()
        _check(lib().meme_seed_batch_host(C.c_void_p(self.h), _p(reads), _p(read_off), C.c_int64(n), C.byref(opt),
                                          C.byref(res)))

        def view(ptr, count, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype, count=count).copy()
        return (view(res.smems, res.total_smems, MEM_TL), view(res.smem_off, n + 1, np.int64),
                view(res.hits, res.total_hits, np.uint64), view(res.hit_off, n + 1, np.int64))

    def seed_batch_resident(self, reads, read_off, opt=None):
        """meme_seed_batch_resident: SMEMs and hits stay in HBM for chain_last_batch_host / extend_last_batch_host; returns the totals."""
        opt = opt or default_seed_opt()
        reads = np.ascontiguousarray(reads, dtype=np.uint8).reshape(-1)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        ts, th = C.c_int64(0), C.c_int64(0)
        _check(lib().meme_seed_batch_resident(C.c_void_p(self.h), _p(reads), _p(read_off), C.c_int64(read_off.shape[0] - 1), C.byref(opt),
                                              C.byref(ts), C.byref(th)))
        return ts.value, th.value

    def seed_batch_resident_ascii(self, reads_ascii, read_off, opt=None):
        """meme_seed_batch_resident_ascii: like seed_batch_resident, for FASTQ letters (converted to codes on the device)."""
        opt = opt or default_seed_opt()
        reads = np.ascontiguousarray(reads_ascii, dtype=np.uint8).reshape(-1)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        ts, th = C.c_int64(0), C.c_int64(0)
        _check(lib().meme_seed_batch_resident_ascii(C.c_void_p(self.h), _p(reads), _p(read_off), C.c_int64(read_off.shape[0] - 1), C.byref(opt),
                                                    C.byref(ts), C.byref(th)))
        return ts.value, th.value

    def seed_reserve(self, nreads, total_bases):
        _check(lib().meme_seed_reserve(C.c_void_p(self.h), C.c_int64(nreads), C.c_int64(total_bases)))

    def chain_last_batch_host(self, contigs, opt):
        """meme_chain_last_batch_host on the batch the last seed_batch_host call seeded.  contigs: (offset, len, is_alt) tuples.
        Returns a dict of numpy arrays (copies of the ctx's pinned buffers)."""
        arr = (Contig * len(contigs))(*[Contig(int(o), int(l), int(a)) for o, l, a in contigs])
        res = ChainHostResult()
        _check(lib().meme_chain_last_batch_host(C.c_void_p(self.h), arr, C.c_int32(len(contigs)), C.byref(opt), C.byref(res)))
        return self._chain_result(res)

    def extend_last_batch_host(self, contigs, chain_opt, ext_opt=None):
        """meme_extend_last_batch_host on the batch the last seed_batch_host call seeded: chaining + seed extension on the device.
        Returns {"reg_off", "regs" (ALNREG records, copies), stats...}."""
        ext_opt = ext_opt or default_ext_opt()
        arr = (Contig * len(contigs))(*[Contig(int(o), int(l), int(a)) for o, l, a in contigs])
        res = ExtHostResult()
        _check(lib().meme_extend_last_batch_host(C.c_void_p(self.h), arr, C.c_int32(len(contigs)), C.byref(chain_opt), C.byref(ext_opt), C.byref(res)))
        n = res.nreads

        def view(ptr, count, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype, count=count).copy()
        return {"reg_off": view(res.reg_off, n + 1, np.int64), "regs": view(res.regs, res.total_regs, ALNREG), "total_chains": int(res.total_chains),
                "n_pairs": int(res.n_pairs), "n_retried": int(res.n_retried), "n_bsw_calls": int(res.n_bsw_calls), "n_tier2": int(res.n_tier2),
                "chain_ms": float(res.chain_ms), "ext_ms": float(res.ext_ms), "bsw_ms": float(res.bsw_ms),
                "n_flt_jobs": int(res.n_flt_jobs), "n_flt_dropped": int(res.n_flt_dropped), "n_exact_prefix": int(res.n_exact_prefix),
                "census_band_cells": int(res.census_band_cells), "census_class": [int(x) for x in res.census_class], "total_seeds": int(res.total_seeds), "n_ext_seeds": int(res.n_ext_seeds)}

    def global_batch_host(self, jobs, opt=None):
        """meme_global_batch_host: banded global alignments with traceback (ksw_global2) of query spans of the batch's reads against
        spans of the text.  jobs: GJOB records.  Returns (GRES records with cigar_off into `cigars`, cigars uint32, kernel_ms)."""
        opt = opt or default_bsw_opt()
        jobs = np.ascontiguousarray(jobs, dtype=GJOB)
        res = GresHost()
        _check(lib().meme_global_batch_host(C.c_void_p(self.h), _p(jobs), C.c_int64(jobs.shape[0]), C.byref(opt), C.byref(res)))

        def view(ptr, count, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype, count=count).copy()
        return view(res.res, res.njobs, GRES), view(res.cigars, res.total_ops, np.uint32), float(res.kernel_ms)

    def gen_cigar_batch_host(self, jobs, opt=None):
        """meme_gen_cigar_batch_host: bwa_gen_cigar2 whole for a batch of its calls (CJOB records: query span of a resident read, text span,
        w_).  Returns (CRES records, cigars uint32, MD bytes (NUL-terminated strings at md_off), kernel_ms)."""
        opt = opt or default_bsw_opt()
        jobs = np.ascontiguousarray(jobs, dtype=CJOB)
        res = CresHost()
        _check(lib().meme_gen_cigar_batch_host(C.c_void_p(self.h), _p(jobs), C.c_int64(jobs.shape[0]), C.byref(opt), C.byref(res)))

        def view(ptr, count, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype, count=count).copy()
        return view(res.res, res.njobs, CRES), view(res.cigars, res.total_ops, np.uint32), view(res.md, res.md_bytes, np.uint8), float(res.kernel_ms)

    def sam_stage_text(self, names, quals=None):
        """meme_sam_stage_text: names (list of bytes, one per read of the resident batch) and qualities (one bytes object laid out like the
        reads' bases, or None)."""
        blob = np.frombuffer(b"".join(names), dtype=np.uint8).copy() if names else np.zeros(1, np.uint8)
        if blob.shape[0] == 0:
            blob = np.zeros(1, np.uint8)
        off = np.zeros(len(names) + 1, np.int64)
        off[1:] = np.cumsum([len(x) for x in names])
        q = np.frombuffer(quals, dtype=np.uint8).copy() if quals is not None else None
        _check(lib().meme_sam_stage_text(C.c_void_p(self.h), _p(blob), _p(off), _p(q) if q is not None else C.c_void_p(0)))

    def sam_format_batch_host(self, recs, blob, contig_names, softclip=0, rg_id=b""):
        """meme_sam_format_batch_host: SAM text of the records (SAM_REC).  Returns (text bytes, text_off int64 array, kernel_ms)."""
        recs = np.ascontiguousarray(recs, dtype=SAM_REC)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        cn = b"".join(n.encode() if isinstance(n, str) else n for n in contig_names)
        cb = np.frombuffer(cn, dtype=np.uint8).copy() if cn else np.zeros(1, np.uint8)
        co = np.zeros(len(contig_names) + 1, np.int32)
        co[1:] = np.cumsum([len(n) for n in contig_names])
        res = SamHost()
        _check(lib().meme_sam_format_batch_host(C.c_void_p(self.h), _p(recs), C.c_int64(recs.shape[0]), _p(blob), C.c_int64(blob.shape[0]), _p(cb), _p(co),
                                                 C.c_int32(len(contig_names)), C.c_int32(int(softclip)), C.c_char_p(rg_id), C.byref(res)))
        if res.nrecs == 0:
            return b"", np.zeros(1, np.int64), 0.0
        off = np.frombuffer((C.c_char * ((res.nrecs + 1) * 8)).from_address(res.text_off), dtype=np.int64).copy()
        text = bytes((C.c_char * res.text_bytes).from_address(res.text)) if res.text_bytes else b""
        return text, off, float(res.kernel_ms)

    def kswv_batch_host(self, jobs, ref, qer, opt=None):
        """meme_kswv_batch_host: the mate-rescue Smith-Waterman batch (mem_sam_pe_batch with the kswv kernels).  jobs: KSWV_JOB records over the
        byte arrays ref / qer.  Returns (KSWR records, kernel_ms)."""
        opt = opt or default_bsw_opt()
        jobs = np.ascontiguousarray(jobs, dtype=KSWV_JOB)
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        qer = np.ascontiguousarray(qer, dtype=np.uint8)
        res = KswvHost()
        _check(lib().meme_kswv_batch_host(C.c_void_p(self.h), _p(jobs), C.c_int64(jobs.shape[0]), _p(ref), C.c_int64(ref.shape[0]), _p(qer),
                                          C.c_int64(qer.shape[0]), C.byref(opt), C.byref(res)))
        if res.njobs == 0:
            return np.zeros(0, KSWR), float(res.kernel_ms)
        buf = (C.c_char * (res.njobs * KSWR.itemsize)).from_address(res.res)
        return np.frombuffer(buf, dtype=KSWR, count=res.njobs).copy(), float(res.kernel_ms)

    def matesw_batch_host(self, regs, reg_off, pes, contigs, l_pac, a=1, b=4, o_del=6, e_del=1, o_ins=6, e_ins=1, pen_unpaired=17, max_matesw=50, min_seed_len=19,
                          batch_reads=512, reads_of=None, first_read=0):
        """meme_matesw_batch_host: mate rescue whole for the pairs of the batch resident on the ctx -- the posing step (mem_sam_pe_batch_pre) and the
        Smith-Waterman jobs (mem_sam_pe_batch).  regs: MATE_REG records of all reads, reg_off per read; pes: 4 x (low, high, failed).
        Returns dict(gar, gar_off, job_off, jobs KSWV_JOB, res KSWR, pose_ms, kernel_ms)."""
        regs = np.ascontiguousarray(regs, dtype=MATE_REG)
        reg_off = np.ascontiguousarray(reg_off, dtype=np.int64)
        pes4 = np.zeros((4, 4), np.int32)
        pes4[:, :3] = np.asarray(pes, np.int32).reshape(4, 3)
        arr = (Contig * len(contigs))(*[Contig(int(o), int(l), int(al)) for o, l, al in contigs])
        opt = MateOpt(a, b, o_del, e_del, o_ins, e_ins, pen_unpaired, max_matesw, min_seed_len, batch_reads)
        res = MateHost()
        _check(lib().meme_matesw_batch_host(C.c_void_p(self.h), C.c_void_p(reads_of.h if reads_of is not None else None), _p(regs), _p(reg_off), C.c_int64(first_read), C.c_int64(reg_off.shape[0] - 1), _p(pes4), arr, C.c_int32(len(contigs)), C.c_int64(l_pac),
                                            C.byref(opt), C.byref(res)))

        def view(ptr, count, dtype):
            if count == 0 or not ptr:
                return np.zeros(0, dtype=dtype)
            buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype, count=count).copy()
        return {"gar": view(res.gar, res.n_gar, np.int32), "gar_off": view(res.gar_off, res.nbatches + 1, np.int64), "job_off": view(res.job_off, res.nbatches + 1, np.int64),
                "jobs": view(res.jobs, res.njobs, KSWV_JOB), "res": view(res.res, res.njobs, KSWR), "pose_ms": float(res.pose_ms), "kernel_ms": float(res.kernel_ms)}

    def chain_batch_host(self, smems, smem_off, hits, hit_off, read_len, contigs, opt):
        """meme_chain_batch_host: chains of seeds the caller brings (numpy arrays laid out as seed_batch_host returns them)."""
        arr = (Contig * len(contigs))(*[Contig(int(o), int(l), int(a)) for o, l, a in contigs])
        smems = np.ascontiguousarray(smems, dtype=MEM_TL)
        smem_off = np.ascontiguousarray(smem_off, dtype=np.int64)
        hits = np.ascontiguousarray(hits, dtype=np.uint64)
        hit_off = np.ascontiguousarray(hit_off, dtype=np.int64)
        read_len = np.ascontiguousarray(read_len, dtype=np.int32)
        res = ChainHostResult()
        _check(lib().meme_chain_batch_host(C.c_void_p(self.h), _p(smems), _p(smem_off), _p(hits), _p(hit_off), _p(read_len), C.c_int64(read_len.shape[0]),
                                           arr, C.c_int32(len(contigs)), C.byref(opt), C.byref(res)))
        return self._chain_result(res)

    @staticmethod
    def _chain_result(res):
        n = res.nreads

        def view(ptr, count, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype, count=count).copy()
        return {"chain_off": view(res.chain_off, n + 1, np.int64), "chains": view(res.chains, res.total_chains, CHAIN),
                "seed_off": view(res.seed_off, n + 1, np.int64), "seeds": view(res.seeds, res.total_seeds, CHAIN_SEED),
                "tree_size": view(res.tree_size, n, np.int32), "frac_rep": view(res.frac_rep, n, np.float32),
                "fallback": view(res.fallback, n, np.uint8), "n_fallback": int(res.n_fallback), "n_tier2": int(res.n_tier2)}

    def seed_batch_device(self, d_reads_ptr, d_read_off_ptr, nreads, total_bases, opt=None) -> SeedResult:
        opt = opt or default_seed_opt()
        res = SeedResult()
        _check(lib().meme_seed_batch_device(C.c_void_p(self.h), C.c_void_p(d_reads_ptr), C.c_void_p(d_read_off_ptr),
                                            C.c_int64(nreads), C.c_int64(total_bases), C.byref(opt), C.byref(res)))
        return res

    # ---- banded SW -------------------------------------------------------------------------------------------
    def bsw_batch(self, pairs, ref, qer, w, opt=None):
        opt = opt or default_bsw_opt()
        assert pairs.dtype == SEQPAIR and pairs.flags.c_contiguous
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        qer = np.ascontiguousarray(qer, dtype=np.uint8)
        _check(lib().meme_bsw_batch(C.c_void_p(self.h), _p(pairs), _p(ref), C.c_int64(ref.shape[0]), _p(qer),
                                    C.c_int64(qer.shape[0]), C.c_int32(pairs.shape[0]), C.c_int32(w), C.byref(opt)))
        return pairs

    def bsw_batch_device(self, d_pairs_ptr, d_ref_ptr, d_qer_ptr, npairs, w, opt=None):
        opt = opt or default_bsw_opt()
        _check(lib().meme_bsw_batch_device(C.c_void_p(self.h), C.c_void_p(d_pairs_ptr), C.c_void_p(d_ref_ptr),
                                           C.c_void_p(d_qer_ptr), C.c_int32(npairs), C.c_int32(w), C.byref(opt)))


def smems_to_slots(smems, smem_off, hits, hit_off, smem_cap=None):
    """Re-shape the flat batch output into the per-read layout the oracle's dump formatter takes."""
    n = smem_off.shape[0] - 1
    counts = np.diff(smem_off)
    cap = int(smem_cap or max(1, counts.max() if n else 1))
    out = np.zeros((n, cap), dtype=MEM_TL)
    hl = []
    for r in range(n):
        k = int(counts[r])
        out[r, :k] = smems[smem_off[r]:smem_off[r + 1]]
        hl.append(hits[hit_off[r]:hit_off[r + 1]])
    return out, counts.astype(np.int32), hl


def stage_entries_torch(ctx, n, d_text, d_pos5):
    """text0123 bytes + 5-byte position image (torch uint8 tensors in HBM) -> 2-bit text words and probe-ready entries."""
    import torch
    L = lib()
    dev = d_text.device
    torch.cuda.synchronize(dev)     # the images were filled on torch's stream; the staging kernels run on the ctx's own stream
    d_pac = torch.empty(L.meme_index_pac64_words(n), dtype=torch.int64, device=dev)
    d_ent = torch.empty(2 * n, dtype=torch.int64, device=dev)
    h = C.c_void_p(ctx.h)
    _check(L.meme_stage_pack_text(h, C.c_void_p(d_text.data_ptr()), C.c_int64(n), C.c_void_p(d_pac.data_ptr())))
    _check(L.meme_stage_build_entries(h, C.c_void_p(d_pos5.data_ptr()), C.c_int64(n), C.c_void_p(d_pac.data_ptr()),
                                      C.c_void_p(d_ent.data_ptr())))
    ctx.sync()
    return d_pac, d_ent


def train_prmi_device(ctx, d_ent, n, bits, partial_threshold=1000):
    """P-RMI trained on the GPU from the staged entries (meme_prmi_train_device).  Returns (d_l2_24, n_l2, d_l1_24, n_l1):
    torch uint8 tensors holding the records in the 24-byte file layout."""
    import torch
    L = lib()
    dev = d_ent.device
    n_l2 = 1 << bits
    d_l2 = torch.empty(n_l2 * 24, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    cnt = C.c_int64(0)
    args = (C.c_void_p(ctx.h), C.c_void_p(d_ent.data_ptr()), C.c_int64(n), C.c_int(bits), C.c_int(partial_threshold), C.c_void_p(d_l2.data_ptr()))
    rc = L.meme_prmi_train_device(*args, C.c_void_p(0), C.c_int64(0), C.byref(cnt))
    n_l1 = int(cnt.value)
    d_l1 = torch.empty(max(n_l1, 1) * 24, dtype=torch.uint8, device=dev)
    if rc != 0:
        if n_l1 == 0:
            _check(rc)
        torch.cuda.synchronize(dev)
        _check(L.meme_prmi_train_device(*args, C.c_void_p(d_l1.data_ptr()), C.c_int64(n_l1), C.byref(cnt)))
    ctx.sync()
    return d_l2, n_l2, d_l1, n_l1


def attach_index_torch(ctx, n, d_pac, d_ent, d_l2_24, n_l2, d_l1_24, n_l1):
    """24-byte model records -> 32-byte records, and the whole index attached to the ctx.  Returns the record tensors."""
    import torch
    L = lib()
    dev = d_ent.device
    torch.cuda.synchronize(dev)
    d_l2 = torch.empty(n_l2 * 32, dtype=torch.uint8, device=dev)
    d_l1 = torch.empty(max(n_l1, 1) * 32, dtype=torch.uint8, device=dev)
    h = C.c_void_p(ctx.h)
    _check(L.meme_stage_rmi32(h, C.c_void_p(d_l2_24.data_ptr()), C.c_int64(n_l2), C.c_void_p(d_l2.data_ptr())))
    _check(L.meme_stage_rmi32(h, C.c_void_p(d_l1_24.data_ptr()), C.c_int64(n_l1), C.c_void_p(d_l1.data_ptr())))
    ctx.sync()
    ctx.attach_index(IndexArrays(n, d_ent.data_ptr(), d_pac.data_ptr(), d_l2.data_ptr(), n_l2, d_l1.data_ptr(), n_l1))
    return d_l2, d_l1


def stage_index_torch(ctx, n, d_text, d_pos5, d_l2_24, n_l2, d_l1_24, n_l1):
    """Multi-GPU / bench start-up: the raw images (text0123 bytes, 5-byte position image, 24-byte P-RMI records) are
    already in HBM as torch uint8 tensors (uploaded, or received through an RCCL broadcast); run the staging kernels
    into torch-owned buffers and attach them.  Returns the tensors that must outlive the ctx."""
    d_pac, d_ent = stage_entries_torch(ctx, n, d_text, d_pos5)
    d_l2, d_l1 = attach_index_torch(ctx, n, d_pac, d_ent, d_l2_24, n_l2, d_l1_24, n_l1)
    return d_pac, d_ent, d_l2, d_l1


def pos5_from_sa_torch(ctx, d_sa, n):
    """u64 suffix array in HBM -> the 5-byte .pos_packed image (torch uint8 tensor)."""
    import torch
    L = lib()
    d_pos5 = torch.zeros(L.meme_index_pos5_bytes(n), dtype=torch.uint8, device=d_sa.device)
    torch.cuda.synchronize(d_sa.device)     # (the zero fill and the upload of d_sa ran on torch's stream, the kernel runs on the ctx's)
    _check(L.meme_stage_pos5_from_sa(C.c_void_p(ctx.h), C.c_void_p(d_sa.data_ptr()), C.c_int64(n),
                                     C.c_void_p(d_pos5.data_ptr())))
    ctx.sync()
    return d_pos5


def fwd_rc_text(fwd: np.ndarray) -> np.ndarray:
    """The reference's .0123 image: the forward strand followed by its reverse complement (codes 0..3)."""
    fwd = np.ascontiguousarray(fwd, dtype=np.uint8)
    return np.concatenate([fwd, (3 - fwd[::-1]).astype(np.uint8)])


def build_sa_device(ctx, text: np.ndarray):
    """Suffix array of the fwd+rc text on the GPU (meme_sa_build_device).  Returns the torch int64 tensor in HBM."""
    import torch
    dev = torch.device("cuda", ctx.device)
    n = int(text.shape[0])
    d_text = torch.from_numpy(text).to(dev)
    d_sa = torch.empty(n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    _check(lib().meme_sa_build_device(C.c_void_p(ctx.h), C.c_void_p(d_text.data_ptr()), C.c_int64(n),
                                      C.c_void_p(d_sa.data_ptr())))
    ctx.sync()
    return d_text, d_sa
