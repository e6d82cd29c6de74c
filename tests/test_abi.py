"""The C-ABI library loads and exports every symbol include/meme_hip.h declares (CPU only, no compute)."""
import ctypes
import os
import re
import subprocess

import pytest

from pymeme import hipapi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "bwa-meme_amd"), "hip"], check=True)
    return ctypes.CDLL(hipapi.LIB_PATH)


def test_header_symbols_are_exported(built):
    hdr = open(os.path.join(REPO, "include", "meme_hip.h")).read()
    declared = set(re.findall(r"\b(meme_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(hipapi.EXPORTS)
    for name in sorted(declared):
        assert hasattr(built, name), name


def test_struct_sizes_match_reference_abi():
    # mem_tl is 24 bytes (reference src/LearnedIndex_seeding.h:121-127), SeqPair 56 (src/bandedSWA.h:90-99)
    assert hipapi.MEM_TL.itemsize == 24 and hipapi.SEQPAIR.itemsize == 56


def test_fails_loudly_without_a_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hipapi.MemeError):
        hipapi.Context(0)
