"""TEST INFRASTRUCTURE — BCF2 records in pure Python, the checker of csrc/bcf.c.

Two independent pieces, both written from the BCF2 specification (hts-specs, VCFv4.3 section 6.3) rather than from the
library's C:
  encode_record()  the call sequence of the reference's _print_vcf_entry (src/print_vcf.c:160-222, :267-378) over plain
                   Python values, with the typed-value rules of htslib's bcf_enc_* helpers (un-vendored dependency of the
                   reference, version not pinned: README "htslib 1.10" / "1.11")
  decode_record()  a reader of the record layout: fixed fields, typed strings / vectors, per-sample fields
Parity with htslib's bytes themselves is unpinned in this image (no htslib to run)."""
import struct

INT8, INT16, INT32, FLOAT, CHAR = 1, 2, 3, 5, 7
FLT_NAME = ["q20", "qd2", "fs60", "mq40"]  # src/init_param.c:15
CS_STR = ["NA", "+", "-", "NA", "+", "+-", "+", "-", "-", "NA"]  # src/print_vcf.c:58-59
GT_HET = [0, 1, 1, 1, 0, 1, 1, 0, 1, 0]  # src/init_param.c:16
# print_vcf_header appends INFO CX, FILTER fail q20 qd2 fs60 mq40 mac1, FORMAT GT FT GL GQ DP MQ QD MC8 AMQ CS CG CX FS (:712-731);
# PASS is entry 0 of every htslib header
HEADER_KEYS = ["PASS", "CX", "fail", "q20", "qd2", "fs60", "mq40", "mac1", "GT", "FT", "GL", "GQ", "DP", "MQ", "QD", "MC8", "AMQ", "CS", "CG", "FS"]
IDS = {k: i for i, k in enumerate(HEADER_KEYS)}


def _itype(lo, hi):
    if hi <= 127 and lo >= -120:
        return INT8, "b"
    if hi <= 32767 and lo >= -32760:
        return INT16, "h"
    return INT32, "i"


def enc_size(n, t):
    if n >= 15:
        return bytes([15 << 4 | t]) + enc_int1(n)
    return bytes([n << 4 | t])


def enc_int1(x):
    t, f = _itype(x, x)
    return enc_size(1, t) + struct.pack("<" + f, x)


def enc_vint(a):
    if len(a) == 0:
        return enc_size(0, 0)
    if len(a) == 1:
        return enc_int1(a[0])
    t, f = _itype(min(a), max(a))
    return enc_size(len(a), t) + struct.pack("<%d%s" % (len(a), f), *a)


def enc_vfloat(a):
    return enc_size(len(a), FLOAT) + struct.pack("<%df" % len(a), *a)


def enc_vchar(s):
    return enc_size(len(s), CHAR) + s


def encode_record(r, rid, rs=b""):
    """r: dict(pos, gt, flt, phred, alt (bytes, 0-2), ref (1 byte), cx_ref (5), cx_gt (5), cg (1 byte), gt_enc, dp, mq, qd, fs,
    gl [floats], counts [8], qual [8])."""
    sh = enc_size(len(rs), CHAR) + rs  # ID (:166-170)
    sh += enc_vchar(r["ref"])
    n_allele = 1
    for a in r["alt"]:
        n_allele += 1
        sh += enc_vchar(bytes([a]))
    flt = r["flt"]
    fid = IDS["PASS"] if not flt else (IDS["mac1"] if flt & 128 else IDS["fail"])
    sh += enc_vint([fid])
    sh += enc_int1(IDS["CX"]) + enc_vchar(r["cx_ref"])
    ind = enc_int1(IDS["GT"]) + enc_vint([r["gt_enc"] >> 4, r["gt_enc"] & 15])
    if flt & 15:  # the copy loop keeps every terminator (:283-296)
        fbuf = b""
        first = True
        for i in range(4):
            if flt >> i & 1:
                if not first:
                    fbuf += b";"
                fbuf += FLT_NAME[i].encode() + b"\0"
                first = False
    else:
        fbuf = b"PASS"
    ind += enc_int1(IDS["FT"]) + enc_size(len(fbuf), CHAR) + fbuf
    for key, v in (("DP", r["dp"]), ("MQ", r["mq"]), ("GQ", r["phred"]), ("QD", r["qd"])):
        ind += enc_int1(IDS[key]) + enc_int1(v)
    ind += enc_int1(IDS["GL"]) + enc_vfloat(r["gl"])
    # the packed record's counts are uint32 (a count beyond 2^32 - 1 saturates); the encoder hands them over as int32, and AMQ lists
    # the classes whose unsigned count is not zero
    cnt = [int(c) & 0xFFFFFFFF for c in r["counts"]]
    ind += enc_int1(IDS["MC8"]) + enc_vint([c - (1 << 32) if c >= 1 << 31 else c for c in cnt])
    n_fmt = 11
    amq = [q for c, q in zip(cnt, r["qual"]) if c > 0]
    if amq:
        ind += enc_int1(IDS["AMQ"]) + enc_vint(amq)
        n_fmt += 1
    cs = CS_STR[r["gt"]].encode()
    ind += enc_int1(IDS["CS"]) + enc_size(len(cs), CHAR) + cs
    ind += enc_int1(IDS["CG"]) + enc_size(1, CHAR) + r["cg"]
    ind += enc_int1(IDS["CX"]) + enc_size(5, CHAR) + r["cx_gt"]
    if GT_HET[r["gt"]]:
        ind += enc_int1(IDS["FS"]) + enc_int1(r["fs"])
        n_fmt += 1
    fixed = struct.pack("<IIiiifII", len(sh) + 24, len(ind), rid, r["pos"] - 1, 1, float(r["phred"]), n_allele << 16 | 1, n_fmt << 24 | 1)
    return fixed + sh + ind


# ---- reader ---------------------------------------------------------------------------------------------------------
def _typed(b, o):
    d = b[o]
    o += 1
    n, t = d >> 4, d & 15
    if n == 15:
        (n,), o = _typed_value(b, o)
    return n, t, o


def _typed_value(b, o):
    n, t, o = _typed(b, o)
    if t == CHAR:
        return b[o : o + n], o + n
    fmt = {INT8: "b", INT16: "h", INT32: "i", FLOAT: "f"}[t] if n else "b"
    w = struct.calcsize(fmt)
    return list(struct.unpack_from("<%d%s" % (n, fmt), b, o)), o + n * w


def decode_record(b):
    l_shared, l_indiv, rid, pos, rlen, qual, nai, nfs = struct.unpack_from("<IIiiifII", b, 0)
    n_allele, n_info, n_fmt, n_sample = nai >> 16, nai & 0xFFFF, nfs >> 24, nfs & 0xFFFFFF
    assert len(b) == 8 + l_shared + l_indiv
    o = 32
    rs, o = _typed_value(b, o)
    alleles = []
    for _ in range(n_allele):
        a, o = _typed_value(b, o)
        alleles.append(a)
    flt, o = _typed_value(b, o)
    info = {}
    for _ in range(n_info):
        (k,), o = _typed_value(b, o)
        v, o = _typed_value(b, o)
        info[HEADER_KEYS[k]] = v
    assert o == 8 + l_shared
    fmt = {}
    order = []
    for _ in range(n_fmt):
        (k,), o = _typed_value(b, o)
        v, o = _typed_value(b, o)  # one sample: the field's vector is the sample's values
        fmt[HEADER_KEYS[k]] = v
        order.append(HEADER_KEYS[k])
    assert o == len(b) and n_sample == 1
    return {"rid": rid, "pos": pos + 1, "rlen": rlen, "qual": qual, "id": rs, "alleles": alleles, "filter": [HEADER_KEYS[i] for i in flt],
            "info": info, "fmt": fmt, "fmt_order": order}
