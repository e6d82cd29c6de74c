"""Host index builder and P-RMI trainer (CPU only)."""
import os

import numpy as np
import pytest

import oracle_py as O
from common import GOLDEN, build_index
from pymeme import synth


def _naive_sa(text):
    n = text.shape[0]
    k = 40
    padded = bytes(text.tolist()) + b"\x03" * (n + k)   # "infinite" T padding
    # shorter-first rule among all-T suffixes == larger position first
    return sorted(range(n), key=lambda i: (padded[i:i + n + k], -i))


def test_suffix_array_matches_naive_sort():
    rng = np.random.default_rng(3)
    fwd = rng.integers(0, 4, size=700, dtype=np.uint8)
    fwd[:6] = 0            # rc ends in TTTTTT
    fwd[100:140] = 3       # long T run
    fwd[300:330] = 0       # long A run -> T run on the reverse strand
    fwd[400:450] = fwd[200:250]
    d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "meme_sa_test")
    os.makedirs(d, exist_ok=True)
    fa = os.path.join(d, "t.fa")
    synth.write_fasta(fa, fwd, contigs=1)
    idx = O.load_index_files(build_index(fa, bits=8))
    text = np.concatenate([fwd, 3 - fwd[::-1]])
    assert np.array_equal(idx.text, text)
    assert idx.sa.tolist() == _naive_sa(text)


def test_index_files_have_reference_layout():
    prefix = build_index(os.path.join(GOLDEN, "g1.fa"))
    l_pac = 48000
    assert os.path.getsize(prefix + ".0123") == 2 * l_pac
    assert os.path.getsize(prefix + ".pos_packed") == 5 * 2 * l_pac
    assert os.path.getsize(prefix + ".suffixarray_uint64") == 8 * (2 * l_pac + 1)
    assert os.path.getsize(prefix + ".pac") == l_pac // 4 + 2
    l2 = os.path.getsize(prefix + ".suffixarray_uint64_L2_PARAMETERS")
    assert l2 == 24 * (1 << 12)
    keys = np.fromfile(prefix + ".suffixarray_uint64", dtype="<u8")
    assert keys[0] == 2 * l_pac
    # keys are non-decreasing in SA order except for the few suffixes that wrap into the padding
    assert (np.diff(keys[1:].astype(np.float64)) < 0).sum() <= 64


def test_prmi_bounds_cover_every_key():
    """Every indexed key must be found inside [pred-lo_err, pred+hi_err] (reference lookup arithmetic,
    src/LearnedIndex_seeding.cpp:186-210, 2145-2146)."""
    prefix = build_index(os.path.join(GOLDEN, "g1.fa"))
    idx = O.load_index_files(prefix)
    n = idx.sa.shape[0]
    l2 = np.fromfile(prefix + ".suffixarray_uint64_L2_PARAMETERS", dtype=[("a", "<f8"), ("b", "<f8"), ("e", "<u8")])
    l1 = np.fromfile(prefix + ".suffixarray_uint64_L1_PARAMETERS", dtype=[("a", "<f8"), ("b", "<f8"), ("e", "<u8")])
    bits = int(np.log2(l2.shape[0]))
    text = np.concatenate([idx.text, np.full(40, 3, np.uint8)])
    rng = np.random.default_rng(0)
    for slot in rng.integers(0, n, size=3000):
        p = int(idx.sa[slot])
        key = 0
        for r in range(32):
            key = (key << 2) | int(text[p + r])
        rec = l2[key >> (64 - bits)]
        pred = rec["a"] + rec["b"] * float(key)
        err = int(rec["e"])
        if err >> 63:
            ps, pn = (err >> 32) & 0x7fffffff, err & 0xffffffff
            rec = l1[ps + int(min(max(pred, 0), pn - 1))]
            pred = rec["a"] + rec["b"] * float(key)
            err = int(rec["e"])
        pos = int(min(max(pred, 0), n - 1))
        lo, hi = (err >> 32) & 0x3fffffff, err & 0x7fffffff
        assert pos - lo <= slot <= pos + hi
