"""csrc/bcf.c (bsc_bcf_record) against the independent BCF2 encoder and reader of oracle/py_bcf.py: random packed records
byte for byte, a hand-assembled record, and the decoded fields against the text line of the same record."""
import ctypes as C

import numpy as np
import pytest

from bs_call_amd import _lib
from bs_call_amd.abi import VCF_REC
from oracle import py_bcf

GT_PAIR = ["AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT"]


def _c_record(rec, rid, rs=b""):
    L = _lib.load()
    ids = _lib.BcfIds()
    L.bsc_bcf_default_ids(C.byref(ids))
    r = np.ascontiguousarray(rec, dtype=VCF_REC).reshape(1)
    need = L.bsc_bcf_record(r.ctypes.data, rid, rs, len(rs), C.byref(ids), None, 0)
    if need <= 0:
        return need
    buf = (C.c_uint8 * need)()
    assert L.bsc_bcf_record(r.ctypes.data, rid, rs, len(rs), C.byref(ids), buf, need) == need
    assert L.bsc_bcf_record(r.ctypes.data, rid, rs, len(rs), C.byref(ids), buf, need - 1) == need  # too small: the size again
    return bytes(buf)


def _as_dict(rec):
    c = rec["core"]
    return dict(pos=int(c["pos"]), gt=int(c["gt"]), flt=int(c["flt"]), phred=int(c["phred"]), alt=c["alt"].tobytes().rstrip(b"\0") if hasattr(c["alt"], "tobytes") else bytes(c["alt"]).rstrip(b"\0"),
                ref=bytes(c["cx_ref"])[2:3], cx_ref=bytes(c["cx_ref"]), cx_gt=bytes(c["cx_gt"]), cg=bytes(c["cg"]), gt_enc=int(c["gt_enc"]),
                dp=int(c["dp"]), mq=int(rec["mq"]), qd=int(c["qd"]), fs=int(c["fs"]), gl=[float(v) for v in c["gl"][: int(c["n_gl"])]],
                counts=[int(v) for v in rec["counts"]], qual=[int(v) for v in rec["qual"]])


def _random_rec(rng):
    r = np.zeros(1, dtype=VCF_REC)[0]
    c = r["core"]
    c["pos"] = int(rng.integers(1, 2**31))
    c["emit"] = 1
    c["gt"] = int(rng.integers(0, 10))
    c["flt"] = int(rng.choice([0, 0, 1, 2, 3, 5, 8, 15, 128, 4, 9]))
    c["phred"] = int(rng.integers(0, 256))
    c["n_gl"] = int(rng.choice([1, 2, 3, 5]))
    c["cg"] = bytes([rng.choice(list(b"CHN?."))])
    alts = [b"", b"A", b"CT"][int(rng.integers(0, 3))]
    c["alt"] = alts
    c["cx_ref"] = bytes(rng.choice(list(b"ACGTN"), 5).tolist())
    c["cx_gt"] = bytes(rng.choice(list(b"ACGTNRYMK"), 5).tolist())
    c["gt_enc"] = int(rng.choice([0x22, 0x24, 0x44, 0x46, 0x48]))
    c["fs"] = int(rng.integers(0, 400))
    c["qd"] = int(rng.integers(0, 300))
    c["dp"] = int(rng.choice([0, 5, 127, 128, 300, 32767, 32768, 100000]))
    c["gl"] = rng.normal(-20, 30, 6).astype(np.float32)
    scale = int(rng.choice([1, 10, 130, 33000, 70000]))
    r["counts"] = rng.integers(0, scale + 1, 8) * (rng.random(8) < 0.7)
    r["qual"] = rng.integers(0, 44, 8)
    r["mq"] = int(rng.integers(0, 256))
    return r


def test_random_records_equal_the_python_encoder_and_decode():
    rng = np.random.default_rng(20261003)
    seen_big = False
    for trial in range(400):
        r = _random_rec(rng)
        rs = [b"", b"rs12345", b"rs1234567\0"][trial % 3]  # an odd number of digits carries its filler byte (bsc_dbsnp_name)
        got = _c_record(r, trial % 25, rs)
        d = _as_dict(r)
        exp = py_bcf.encode_record(d, trial % 25, rs)
        assert got == exp, trial
        dec = py_bcf.decode_record(got)
        assert dec["pos"] == d["pos"] and dec["rid"] == trial % 25 and dec["qual"] == float(d["phred"]) and dec["id"] == rs
        assert dec["alleles"] == [d["ref"]] + [bytes([a]) for a in d["alt"]]
        assert dec["filter"] == (["PASS"] if d["flt"] == 0 else (["mac1"] if d["flt"] & 128 else ["fail"]))
        assert dec["info"] == {"CX": d["cx_ref"]}
        f = dec["fmt"]
        assert f["GT"] == [d["gt_enc"] >> 4, d["gt_enc"] & 15] and f["DP"] == [d["dp"]] and f["MQ"] == [d["mq"]] and f["GQ"] == [d["phred"]]
        assert f["QD"] == [d["qd"]] and f["MC8"] == d["counts"] and f["CG"] == d["cg"] and f["CX"] == d["cx_gt"]
        assert np.array(f["GL"], dtype=np.float32).tobytes() == np.array(d["gl"], dtype=np.float32).tobytes()
        assert f["CS"].decode() == py_bcf.CS_STR[d["gt"]]
        assert ("FS" in f) == bool(py_bcf.GT_HET[d["gt"]]) and ("AMQ" in f) == any(d["counts"])
        assert dec["fmt_order"][:8] == ["GT", "FT", "DP", "MQ", "GQ", "QD", "GL", "MC8"]
        names = [n for i, n in enumerate(py_bcf.FLT_NAME) if d["flt"] >> i & 1]
        assert f["FT"] == (b"".join((b";" if k else b"") + n.encode() + b"\0" for k, n in enumerate(names)) if names else b"PASS")
        seen_big |= max(d["counts"]) > 32767
    assert seen_big


def test_a_record_assembled_by_hand():
    """chr index 3, position 1000, a heterozygous CT call at a C reference (ALT T), PASS, phred 50, no dbSNP name.
    Typed values by the BCF2 rules: 0x07 = empty string; 0x17 c = one character; 0x11 v = one int8; 0x57 = five characters;
    0x21 = two int8; 0x15 = one float; 0x81 = eight int8; 0x47 "PASS"."""
    r = np.zeros(1, dtype=VCF_REC)[0]
    c = r["core"]
    c["pos"], c["emit"], c["gt"], c["flt"], c["phred"], c["n_gl"], c["cg"] = 1000, 1, 6, 0, 50, 3, b"H"
    c["alt"], c["cx_ref"], c["cx_gt"], c["gt_enc"], c["fs"], c["qd"], c["dp"] = b"T", b"AACGT", b"AAYGT", 0x24, 3, 2, 20
    c["gl"][:3] = [-5.0, 0.0, -7.5]
    r["counts"] = [0, 9, 0, 11, 0, 4, 0, 6]
    r["qual"] = [0, 30, 0, 31, 0, 32, 0, 33]
    r["mq"] = 60
    shared = bytes([0x07, 0x17]) + b"C" + bytes([0x17]) + b"T" + bytes([0x11, 0]) + bytes([0x11, 1, 0x57]) + b"AACGT"
    import struct

    indiv = (bytes([0x11, 8, 0x21, 2, 4]) + bytes([0x11, 9, 0x47]) + b"PASS" + bytes([0x11, 12, 0x11, 20]) + bytes([0x11, 13, 0x11, 60])
             + bytes([0x11, 11, 0x11, 50]) + bytes([0x11, 14, 0x11, 2]) + bytes([0x11, 10, 0x35]) + struct.pack("<3f", -5.0, 0.0, -7.5)
             + bytes([0x11, 15, 0x81, 0, 9, 0, 11, 0, 4, 0, 6]) + bytes([0x11, 16, 0x41, 30, 31, 32, 33]) + bytes([0x11, 17, 0x17]) + b"+"
             + bytes([0x11, 18, 0x17]) + b"H" + bytes([0x11, 1, 0x57]) + b"AAYGT" + bytes([0x11, 19, 0x11, 3]))
    fixed = struct.pack("<IIiiifII", len(shared) + 24, len(indiv), 3, 999, 1, 50.0, 2 << 16 | 1, 13 << 24 | 1)
    assert _c_record(r, 3) == fixed + shared + indiv


def test_not_written_and_bad_arguments():
    r = np.zeros(1, dtype=VCF_REC)[0]
    assert _c_record(r, 0) == 0  # emit == 0
    r["core"]["emit"], r["core"]["gt"] = 1, 11
    assert _c_record(r, 0) == -1
    L = _lib.load()
    assert L.bsc_bcf_record(None, 0, None, 0, None, None, 0) == -1
