"""ctypes bindings of libmeme_host.so (host index construction; see host/meme_host_capi.cpp)."""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(PKG, "libmeme_host.so")
_lib = None

RMI_DTYPE = np.dtype([("icpt", "<f8"), ("slope", "<f8"), ("err", "<u8")])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s missing: run `make -C bwa-meme_amd host`" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.meme_host_free.argtypes = [C.c_void_p]
        _lib.meme_host_free.restype = None
    return _lib


def build_sa(fwd: np.ndarray, threads=0):
    fwd = np.ascontiguousarray(fwd, dtype=np.uint8)
    l_pac = fwd.shape[0]
    text = np.empty(2 * l_pac, dtype=np.uint8)
    sa = np.empty(2 * l_pac, dtype=np.uint64)
    rc = lib().meme_host_build_sa(C.c_void_p(fwd.ctypes.data), C.c_int64(l_pac), C.c_void_p(text.ctypes.data),
                                  C.c_void_p(sa.ctypes.data), C.c_int(threads))
    if rc:
        raise RuntimeError("meme_host_build_sa failed (%d)" % rc)
    return text, sa


def train_prmi(text: np.ndarray, sa: np.ndarray, bits=0, partial_threshold=1000, threads=0):
    n = text.shape[0]
    l2, l1 = C.c_void_p(), C.c_void_p()
    n2, n1 = C.c_int64(), C.c_int64()
    rc = lib().meme_host_train_prmi(C.c_void_p(text.ctypes.data), C.c_int64(n), C.c_void_p(sa.ctypes.data),
                                    C.c_int(bits), C.c_int(partial_threshold), C.c_int(threads), C.byref(l2),
                                    C.byref(n2), C.byref(l1), C.byref(n1))
    if rc:
        raise RuntimeError("meme_host_train_prmi failed (%d)" % rc)
    a2 = np.frombuffer((C.c_uint8 * (n2.value * 24)).from_address(l2.value), dtype=RMI_DTYPE).copy()
    a1 = np.frombuffer((C.c_uint8 * (max(n1.value, 1) * 24)).from_address(l1.value), dtype=RMI_DTYPE).copy()[:n1.value]
    lib().meme_host_free(l2)
    lib().meme_host_free(l1)
    return a1, a2


def write_index(prefix: str, fwd, text, sa, l1, l2, n_contigs=1, with_keys=False):
    l1c = np.ascontiguousarray(l1)
    l2c = np.ascontiguousarray(l2)
    rc = lib().meme_host_write_index(prefix.encode(), C.c_void_p(fwd.ctypes.data), C.c_int64(fwd.shape[0]),
                                     C.c_void_p(text.ctypes.data), C.c_void_p(sa.ctypes.data),
                                     C.c_void_p(l1c.ctypes.data if l1c.shape[0] else None), C.c_int64(l1c.shape[0]),
                                     C.c_void_p(l2c.ctypes.data), C.c_int64(l2c.shape[0]), C.c_int(n_contigs),
                                     C.c_int(1 if with_keys else 0))
    if rc:
        raise RuntimeError("meme_host_write_index failed (%d)" % rc)
