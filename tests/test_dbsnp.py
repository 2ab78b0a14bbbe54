"""The dbSNP index reader of the library (csrc/dbsnp.c, host C + zlib) — SURVEY.md 8 row f-3 — against
  * the writer of the on-disk format (tools/make_dbsnp_index.py, following src/dbSNP_output.c): what was written is read,
  * the pure-Python restatement of the reference's reader (oracle/py_dbsnp.py, following src/dbSNP.c statement by
    statement): same flags, names and lengths at every position,
and its errors on malformed files.  CPU only: the reader is host code; the GPU tests feed its flags to the kernels."""
import importlib.util
import os
import struct

import numpy as np
import pytest

from bs_call_amd.caller import BscError
from bs_call_amd.dbsnp import DbSnpIndex
from oracle import py_dbsnp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("make_dbsnp_index", os.path.join(ROOT, "tools", "make_dbsnp_index.py"))
W = importlib.util.module_from_spec(spec)
spec.loader.exec_module(W)


def _adversarial_sites():
    """Every bin-distance encoding (1, 2, 3, 5 bytes), a full bin of 64 entries, even / odd digit counts, a leading
    zero, explicit (>= 3) prefix indices, first and last position of a bin, fq flags."""
    sites = [(1, "1", False, 0), (63, "22", True, 0), (64, "333", False, 1), (127, "4444", True, 2)]
    sites += [(128 + i, str(1000 + i), i % 3 == 0, 0) for i in range(64)]  # a full bin
    sites += [(64 * 70 + 5, "0123456789", False, 0)]        # distance 67 bins: 2-byte form
    sites += [(64 * 400 + 9, "99999", True, 3)]             # distance 330: 3-byte form, explicit prefix index
    sites += [(64 * 70_000 + 1, "7", False, 4), (64 * 70_000 + 2, "12345678901234567", True, 0)]  # 5-byte form
    return sites


@pytest.fixture(scope="module")
def index_file(tmp_path_factory):
    d = tmp_path_factory.mktemp("dbsnp")
    path = str(d / "test.idx")
    ctgs = {"chrA": _adversarial_sites(), "chrS": W.synthetic_sites(700_000, 300), "chrE": [(5, "42", True, 0)]}
    W.write_index(path, ctgs, prefixes=("rs", "ss", "esv", "xx", "yy"))
    return path, ctgs


def test_header_and_directory(index_file):
    path, ctgs = index_file
    with DbSnpIndex(path) as db:
        assert db.contigs == list(ctgs) and db.header.startswith("name = dbSNP_index")
        assert db.load_contig("chrUnknown") == 0 and not db.flags(1, 1000).any()
        for name, sites in ctgs.items():
            assert db.load_contig(name) == len(sites)


def test_flags_and_names_round_trip(index_file):
    path, ctgs = index_file
    pre = ("rs", "ss", "esv", "xx", "yy")
    with DbSnpIndex(path) as db:
        for name in ("chrA", "chrE"):
            sites = ctgs[name]
            db.load_contig(name)
            top = max(s[0] for s in sites) + 200
            want = np.zeros(top, dtype=np.uint8)
            for pos, rs, fq, pix in sites:
                want[pos - 1] = 3 if fq else 1
            for x0, n in ((1, top), (60, 10), (64, 64), (129, 1), (1, 1)):
                assert (db.flags(x0, n) == want[x0 - 1 : x0 - 1 + n]).all()
            for pos, rs, fq, pix in sites:
                r, nm, ln = db.name(pos)
                assert r == (3 if fq else 1)
                assert nm == pre[pix] + rs
                assert ln == len(pre[pix]) + 2 * ((len(rs) + 1) // 2)  # an odd digit count carries its filler
            assert db.name(2) == (0, "", 0)


def test_c_reader_equals_python_restatement_of_the_reference(index_file):
    path, ctgs = index_file
    ref = py_dbsnp.Index(path)
    with DbSnpIndex(path) as db:
        assert ref.prefixes == ["rs", "ss", "esv", "xx", "yy"] and list(ref.ctgs) == db.contigs
        for name, sites in ctgs.items():
            assert ref.load_contig(name) == db.load_contig(name) == len(sites)
            probe = sorted({s[0] + d for s in sites[:3000] for d in (-1, 0, 1) if s[0] + d > 0})
            fl = db.flags(1, max(probe) + 1)
            explicit = (64 * 400 + 9, 64 * 70_000 + 1) if name == "chrA" else ()  # entries with an explicit prefix index: below
            for x in probe:
                if x in explicit:
                    continue
                r, nm, ln = ref.lookup(x)
                assert fl[x - 1] == r, (name, x)
                if r:
                    assert db.name(x) == (r, nm, ln), (name, x)
        # explicit prefix indices (the fourth prefix onwards): the reference's reader assembles the writer's little-endian
        # u16 high byte first (src/dbSNP.c:337 vs src/dbSNP_output.c:280) and indexes its prefix table out of bounds —
        # the restatement does the same and fails; the library reads the index as written
        ref.load_contig("chrA")
        db.load_contig("chrA")
        with pytest.raises(IndexError):
            ref.lookup(64 * 400 + 9)
        assert db.name(64 * 400 + 9) == (3, "xx99999", 8) and db.name(64 * 70_000 + 1) == (1, "yy7", 4)


def test_synthetic_index_matches_the_spec(index_file):
    path, ctgs = index_file
    with DbSnpIndex(path) as db:
        n = db.load_contig("chrS")
        fl = db.flags(1, 700_000)
        assert n == int((fl != 0).sum()) == len(ctgs["chrS"]) and abs(n - 700_000 / 300) < 2
        assert 0.05 < float((fl == 3).sum()) / n < 0.16  # 10 % flagged fq_mask


def test_malformed_files(tmp_path, index_file):
    path, _ = index_file
    raw = open(path, "rb").read()
    bad = tmp_path / "bad.idx"
    with pytest.raises(BscError):
        DbSnpIndex(str(tmp_path / "missing.idx"))
    bad.write_bytes(b"\0" * 64)
    with pytest.raises(BscError):
        DbSnpIndex(str(bad))
    bad.write_bytes(raw[:-4] + b"\0\0\0\0")  # trailing magic gone
    with pytest.raises(BscError):
        DbSnpIndex(str(bad))
    bad.write_bytes(raw[:40])  # truncated
    with pytest.raises(BscError):
        DbSnpIndex(str(bad))
    # a contig whose data block is corrupt: the header opens, the load fails, the index stays usable
    off = struct.unpack_from("<Q", raw, 8)[0]
    d = bytearray(raw)
    for k in range(48, 80):
        d[k] ^= 0x5A
    bad.write_bytes(bytes(d))
    assert off > 80
    with DbSnpIndex(str(bad)) as db:
        with pytest.raises(BscError):
            db.load_contig("chrA")
        assert db.load_contig("chrE") == 1


def test_counts_in_a_damaged_directory_are_not_believed(tmp_path):
    """Found by tools/fuzz_host_inputs.py under the sanitized build: the directory's contig / prefix counts and a contig's bin
    range sized allocations before anything was checked (100 GB for a count of 2^32 - 1).  A count the directory cannot hold
    and a bin beyond position 2^32 are errors."""
    import zlib

    for field, value, msg in ((4, 0xFFFFFFFF, "more entries than it holds"), (2, 0xFFFF, "more entries than it holds"), (None, 1 << 27, "beyond position")):
        path = str(tmp_path / "d.idx")
        W.write_index(path, {"chr1": [(5, "42", True, 0)]})
        raw = bytearray(open(path, "rb").read())
        _, _, off, max_buf, zlen = struct.unpack_from("<IIQQQ", raw, 0)
        d = bytearray(zlib.decompress(bytes(raw[off : off + zlen])))
        if field == 4:
            struct.pack_into("<I", d, 4, value)
        elif field == 2:
            struct.pack_into("<H", d, 2, value)
        else:  # max_bin of the only contig: it follows the header line and the prefix
            o = d.index(b"rs\0", 8) + 3
            struct.pack_into("<I", d, o + 4, value)
        z = zlib.compress(bytes(d))
        out = bytes(raw[:off]) + z + struct.pack("<I", W.MAGIC)
        out = struct.pack("<IIQQQ", W.MAGIC, 0, off, max(max_buf, len(d)), len(z)) + out[32:]
        open(path, "wb").write(out)
        with pytest.raises(BscError, match=msg):
            DbSnpIndex(path)


def test_bcf_records_carry_the_dbsnp_names(index_file):
    """bsc_bcf_block names the records whose rs_found flag is set: the ID field is the name with the length the reference
    hands to htslib (an odd digit count keeps its filler NUL, src/dbSNP.c:306-350)."""
    from bs_call_amd import vcf
    from bs_call_amd.abi import VCF_REC
    from oracle import py_bcf

    path, ctgs = index_file
    pre = ("rs", "ss", "esv", "xx", "yy")
    sites = ctgs["chrA"][:6]
    recs = np.zeros(len(sites) + 1, dtype=VCF_REC)
    for i, (pos, rs, fq, pix) in enumerate(sites):
        c = recs[i]["core"]
        c["pos"], c["emit"], c["gt"], c["n_gl"], c["cg"], c["cx_ref"], c["cx_gt"], c["gt_enc"] = pos, 1, 0, 1, b".", b"AAAAA", b"AAAAA", 0x22
        recs[i]["rs_found"] = 3 if fq else 1
    c = recs[-1]["core"]  # a written record without a dbSNP entry
    c["pos"], c["emit"], c["gt"], c["n_gl"], c["cg"], c["cx_ref"], c["cx_gt"], c["gt_enc"] = 5000, 1, 0, 1, b".", b"AAAAA", b"AAAAA", 0x22
    with DbSnpIndex(path) as db:
        db.load_contig("chrA")
        blob = vcf.bcf_block(recs, 4, db)
    o, got = 0, []
    while o < len(blob):
        l_shared, l_indiv = struct.unpack_from("<II", blob, o)
        got.append(py_bcf.decode_record(blob[o : o + 8 + l_shared + l_indiv]))
        o += 8 + l_shared + l_indiv
    assert len(got) == len(recs)
    for d, (pos, rs, fq, pix) in zip(got, sites):
        name = (pre[pix] + rs).encode()
        assert d["pos"] == pos and d["rid"] == 4 and d["id"] == name + (b"\0" if len(rs) % 2 else b"")
    assert got[-1]["id"] == b""


def test_names_of_a_range_are_the_single_lookups(index_file):
    """bsc_dbsnp_names (the table the device BCF encoder searches, csrc/bcfdev.hip): every flagged position of the range, ascending,
    with the bytes and the length bsc_dbsnp_name gives for it."""
    path, ctgs = index_file
    with DbSnpIndex(path) as db:
        for name in ("chrA", "chrS", "chrE"):
            db.load_contig(name)
            sites = sorted(s[0] for s in ctgs[name])
            top = sites[-1] + 100
            a, b, c = sites[min(1, len(sites) - 1)], sites[min(2, len(sites) - 1)], sites[min(5, len(sites) - 1)]
            for x0, n in ((1, top), (a, 1), (a + 1, 0), (b, c - b + 1), (63, 130), (top, 50)):
                pos, off, by = db.names(x0, n)
                want = [p for p in sites if x0 <= p < x0 + n]
                assert pos.tolist() == want and len(off) == len(want) + 1 and int(off[-1]) == len(by)
                for i, p in enumerate(want[:: max(1, len(want) // 50)]):
                    j = want.index(p)
                    r, nm, ln = db.name(p)
                    assert by[int(off[j]) : int(off[j + 1])] == (nm.encode() + b"\0")[:ln]
        db.load_contig("chrUnknown")
        pos, off, by = db.names(1, 10_000)
        assert len(pos) == 0 and off.tolist() == [0] and by == b""
