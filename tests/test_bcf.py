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


def _read_bgzf(path):
    """BGZF reader from the SAM specification: concatenated gzip members, each with a 'BC' extra field."""
    import struct
    import zlib

    raw = open(path, "rb").read()
    out, o = bytearray(), 0
    while o < len(raw):
        assert raw[o : o + 4] == b"\x1f\x8b\x08\x04"
        xlen = struct.unpack_from("<H", raw, o + 10)[0]
        assert raw[o + 12 : o + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", raw, o + 16)[0] + 1
        body = raw[o + 12 + xlen : o + bsize - 8]
        data = zlib.decompress(body, -15)
        crc, isize = struct.unpack_from("<II", raw, o + bsize - 8)
        assert zlib.crc32(data) & 0xFFFFFFFF == crc and isize == len(data)
        out += data
        o += bsize
    assert raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))  # the EOF marker block
    return bytes(out)


def test_bcf_file_round_trip(tmp_path):
    """Header + a few thousand records through write_bcf (BGZF) and back through a reader written from the specifications;
    the same stream uncompressed; gzip's own reader accepts the BGZF file (it is a multi-member gzip file)."""
    import gzip
    import struct

    from bs_call_amd import vcf

    rng = np.random.default_rng(5)
    recs = np.zeros(3000, dtype=VCF_REC)
    for i in range(len(recs)):
        recs[i] = _random_rec(rng)
        recs[i]["core"]["pos"] = 1000 + 3 * i
    recs[17]["core"]["emit"] = 0  # not written
    hdr = vcf.header_text([("chr1", 248956422), ("chr2", 242193529, "GRCh38")], "SAMPLE1", date=(3, 10, 2026))
    assert hdr.startswith("##fileformat=VCFv4.2\n##FILTER=<ID=PASS,") and hdr.endswith("\tFORMAT\tSAMPLE1\n")
    assert "##contig=<ID=chr2,length=242193529,assembly=GRCh38>\n" in hdr and "##source=bs_call_v2.1,under_conversion=0.01,over_conversion=0.05,mapq_thresh=20,bq_thresh=20\n" in hdr
    # the keys appear in the order that gives the dictionary indices of bsc_bcf_default_ids
    keys = []
    for ln in hdr.splitlines():
        if ln.startswith(("##FILTER=<ID=", "##INFO=<ID=", "##FORMAT=<ID=")):
            k = ln.split("<ID=")[1].split(",")[0]
            if k not in keys:
                keys.append(k)
    assert keys == py_bcf.HEADER_KEYS
    blocks = [vcf.bcf_block(recs[:1000], 0), vcf.bcf_block(recs[1000:], 1)]
    p1, p2 = tmp_path / "out.bcf", tmp_path / "plain.bcf"
    vcf.write_bcf(p1, hdr, blocks)
    vcf.write_bcf(p2, hdr, blocks, compressed=False)
    stream = _read_bgzf(p1)
    assert stream == open(p2, "rb").read() == gzip.open(p1, "rb").read()
    assert stream[:5] == b"BCF\x02\x02"
    l_text = struct.unpack_from("<I", stream, 5)[0]
    assert stream[9 : 9 + l_text] == hdr.encode() + b"\0"
    o, k, n = 9 + l_text, 0, 0
    while o < len(stream):
        l_shared, l_indiv = struct.unpack_from("<II", stream, o)
        rec = stream[o : o + 8 + l_shared + l_indiv]
        while not recs[k]["core"]["emit"]:
            k += 1
        d = py_bcf.decode_record(rec)
        assert d["pos"] == int(recs[k]["core"]["pos"]) and d["rid"] == (0 if k < 1000 else 1)
        assert rec == _c_record(recs[k], 0 if k < 1000 else 1)
        o += len(rec)
        k += 1
        n += 1
    assert n == 2999


def test_vcf_text_file(tmp_path):
    """Header + data lines as plain and as bgzip'ed VCF: the same text either way; the lines are bsc_vcf_format_rec's."""
    import gzip

    from bs_call_amd import vcf

    rng = np.random.default_rng(9)
    recs = np.zeros(200, dtype=VCF_REC)
    for i in range(len(recs)):
        recs[i] = _random_rec(rng)
        recs[i]["core"]["pos"] = 10 + i
    hdr = vcf.header_text([("chr1", 1000)], "S", benchmark_mode=True)
    assert "fileDate" not in hdr and "##source" not in hdr  # --benchmark-mode leaves the run-dependent lines out
    lines = vcf.format_records_c(recs, "chr1")
    assert len(lines) == 200 and all(ln.split("\t")[0] == "chr1" and ln.count("\t") == 9 for ln in lines)
    p1, p2 = tmp_path / "o.vcf", tmp_path / "o.vcf.gz"
    vcf.write_vcf(p1, hdr, [lines[:50], [], lines[50:]])
    vcf.write_vcf(p2, hdr, [lines], bgzip=True)
    text = open(p1).read()
    assert text == hdr + "\n".join(lines) + "\n" == gzip.open(p2, "rt").read()
    assert _read_bgzf(p2).decode() == text
