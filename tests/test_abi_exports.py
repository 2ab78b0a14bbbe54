"""CPU checks of the boundary: the C-ABI library loads and exports every symbol include/bscall_amd.h
declares, the record layouts match the reference's sizes, and creating a context without a GPU fails
loudly (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

import bs_call_amd as B
from bs_call_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "bscall_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bsc_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_all_exported():
    L = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 13
    for s in syms:
        assert hasattr(L, s), "libbscall_amd.so does not export %s" % s
    assert sorted(_lib.EXPORTS) == syms  # the binding knows exactly the declared set
    assert L.bsc_abi_version() == 2


def test_record_layouts():
    assert B.PILEUP.itemsize == 104 and B.GT_METH.itemsize == 200 and B.TEMPLATE.itemsize == 40
    assert B.PILEUP.fields["n"][1] == 64 and B.PILEUP.fields["quality"][1] == 68 and B.PILEUP.fields["mapq2"][1] == 100
    f = B.GT_METH.fields
    assert (f["qual"][1], f["gt_prob"][1], f["fisher_strand"][1], f["mq"][1], f["aq"][1], f["max_gt"][1]) == (
        64, 96, 176, 184, 188, 192)
    assert ctypes.sizeof(_lib.Params) == 32 and ctypes.sizeof(_lib.Stats) == 128


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(B.BscError) as ei:
        B.SiteCaller()
    assert ei.value.code == -4 and "no CPU path" in str(ei.value)


def test_product_does_not_touch_oracle():
    """Nothing under bs_call_amd/ or include/ may reference oracle/ (bsmath.h is shared the other way)."""
    bad = []
    for d in ("bs_call_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, d)):
            for fn in fns:
                if fn.endswith((".py", ".c", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, fn), errors="replace").read()
                    if re.search(r"(import\s+oracle|from\s+oracle|oracle/|liboracle|orc_)", txt):
                        if fn == "bsmath.h":  # mentions oracle/ in a comment only
                            continue
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_synth_host_twin_is_deterministic_and_windowed():
    a, ra = B.synth_pileup_host(1234, 1000, 3000, 30)
    b, rb = B.synth_pileup_host(1234, 2000, 1000, 30)
    assert a[1000:2000].tobytes() == b.tobytes() and (ra[1000:2000] == rb).all()
    assert (a["counts"].sum(axis=(1, 2)) == a["n"]).all()
    assert abs(a["n"].mean() - 30) < 1.0
    assert set(np.unique(ra)) <= {1, 2, 3, 4}
    pn, rn = B.synth_pileup_host(1235, 85_000, 20_000, 10, flags=1)  # seed 1235: run 9 (sites 90000..99999) is N
    assert (rn[5000:15000] == 0).all() and (rn[:5000] != 0).all() and (rn[15000:] != 0).all()
    assert (pn["n"][5000:15000] == 0).all()
