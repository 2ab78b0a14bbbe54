#!/usr/bin/env python3
"""Writes BAM files (SAM specification sections 4.1-4.2: BGZF blocks, header, alignment records) from plain Python
records — the test input of csrc/bamio.c and oracle/py_bam.py; there is no samtools / htslib in this image.

write_bam(path, refs, records, text=None, block=0xff00)
  refs     [(name, length), ...]
  records  dicts: name, flag, tid, pos (0-based, -1 = none), mapq, cigar [(op char, length), ...], mtid, mpos, tlen, seq (str over
           =ACMGRSVTWYHKDBN), qual (list of ints; None = 0xff), aux (bytes, already encoded; see aux_* helpers)
A synthetic WGBS generator for end-to-end tests lives in wgbs_records()."""
import struct
import zlib

SEQ_CODES = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
CIGAR_CODES = {c: i for i, c in enumerate("MIDNSHP=X")}
BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def bgzf_block(data: bytes) -> bytes:
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(data) + co.flush()
    total = 18 + len(body) + 8
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", total - 1) + body
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def aux_char(tag, c):
    return tag.encode() + b"A" + c.encode()


def aux_str(tag, s):
    return tag.encode() + b"Z" + s.encode() + b"\0"


def aux_int(tag, v):
    return tag.encode() + b"i" + struct.pack("<i", v)


def encode_record(r) -> bytes:
    name = r["name"].encode() + b"\0"
    seq = r.get("seq", "")
    l_seq = len(seq)
    packed = bytearray((l_seq + 1) // 2)
    for i, c in enumerate(seq):
        packed[i >> 1] |= SEQ_CODES[c] << (4 if i % 2 == 0 else 0)
    qual = r.get("qual")
    qb = bytes([0xFF] * l_seq) if qual is None else bytes(qual)
    cig = b"".join(struct.pack("<I", n << 4 | CIGAR_CODES[op]) for op, n in r.get("cigar", []))
    span = sum(n for op, n in r.get("cigar", []) if op in "MDN=X")
    pos = r.get("pos", -1)
    body = struct.pack("<iiBBHHHIiii", r.get("tid", -1), pos, len(name), r.get("mapq", 0), reg2bin(max(pos, 0), max(pos, 0) + max(span, 1)),
                       len(r.get("cigar", [])), r["flag"], l_seq, r.get("mtid", -1), r.get("mpos", -1), r.get("tlen", 0))
    body += name + cig + bytes(packed) + qb + r.get("aux", b"")
    return struct.pack("<I", len(body)) + body


def write_bam(path, refs, records, text=None, block=0xFF00, aligned=False):
    """aligned: BGZF blocks are cut at record boundaries, as htslib's writer cuts them (bgzf_flush_try) — never through a record"""
    if text is None:
        text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    t = text.encode()
    data = bytearray(b"BAM\1" + struct.pack("<I", len(t)) + t + struct.pack("<i", len(refs)))
    for name, ln in refs:
        data += struct.pack("<I", len(name) + 1) + name.encode() + b"\0" + struct.pack("<I", ln)
    if aligned:
        with open(path, "wb") as f:
            for r in records:
                e = encode_record(r)
                if len(data) and len(data) + len(e) > block:
                    f.write(bgzf_block(bytes(data)))
                    data = bytearray()
                data += e
            if data:
                f.write(bgzf_block(bytes(data)))
            f.write(BGZF_EOF)
        return
    for r in records:
        data += encode_record(r)
    with open(path, "wb") as f:
        for o in range(0, len(data), block):
            f.write(bgzf_block(bytes(data[o : o + block])))
        f.write(BGZF_EOF)


def aux_text(aux: bytes) -> str:
    """The optional fields of encode_record's `aux` bytes (A / i / Z tags, as the aux_* helpers make them) as SAM text."""
    out, o = [], 0
    while o + 3 <= len(aux):
        tag, ty = aux[o : o + 2].decode(), chr(aux[o + 2])
        o += 3
        if ty == "A":
            out.append("%s:A:%s" % (tag, chr(aux[o])))
            o += 1
        elif ty == "i":
            out.append("%s:i:%d" % (tag, struct.unpack_from("<i", aux, o)[0]))
            o += 4
        elif ty == "Z":
            e = aux.index(b"\0", o)
            out.append("%s:Z:%s" % (tag, aux[o:e].decode()))
            o = e + 1
        else:
            raise ValueError("aux_text: type %r not handled" % ty)
    return "\t".join(out)


def write_sam(path, refs, records, text=None, bgzf=False, block=0xFF00):
    """The same records as SAM text (plain, or in BGZF blocks like `samtools view -h | bgzip`)."""
    if text is None:
        text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    lines = [text]
    for r in records:
        name = lambda t: "*" if t < 0 else refs[t][0]
        mt = r.get("mtid", -1)
        rnext = "=" if (mt >= 0 and mt == r.get("tid", -1)) else name(mt)
        cig = "".join("%d%s" % (n, op) for op, n in r.get("cigar", [])) or "*"
        qual = r.get("qual")
        seq = r.get("seq", "") or "*"
        q = "*" if (qual is None or not r.get("seq")) else "".join(chr(33 + v) for v in qual)
        f = [r["name"], str(r["flag"]), name(r.get("tid", -1)), str(r.get("pos", -1) + 1), str(r.get("mapq", 0)), cig, rnext, str(r.get("mpos", -1) + 1),
             str(r.get("tlen", 0)), seq, q]
        a = aux_text(r.get("aux", b""))
        lines.append("\t".join(f) + ("\t" + a if a else "") + "\n")
    data = "".join(lines).encode()
    with open(path, "wb") as f:
        if not bgzf:
            f.write(data)
            return
        for o in range(0, len(data), block):
            f.write(bgzf_block(data[o : o + block]))
        f.write(BGZF_EOF)


def wgbs_records(rng, ref_codes, tid, n_pairs, read_len=100, insert=300, strand_tag="XB", meth_cpg=0.8, conv=120 / 128, err=0.005, het_every=1000):
    """Paired WGBS alignments over one contig (ref_codes: 1..4 = ACGT, 0 = N; position 1 first), coordinate-sorted: the read
    generator of SURVEY.md 8(d) (level L-reads) as BAM records.  The forward read is read 1 on a FORWARD template."""
    import numpy as np

    L = len(ref_codes)
    starts = np.sort(rng.integers(0, max(1, L - insert - 2), n_pairs))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    letters = "NACGT"
    recs = []
    for i, s in enumerate(starts):
        s = int(s)
        c2t = bool(rng.integers(0, 2))
        fwd_first = bool(rng.integers(0, 2))
        mates = []
        for k, off in enumerate((0, insert - read_len)):
            p0 = s + off
            seq = []
            for j in range(read_len):
                g = p0 + j
                code = int(ref_codes[g]) if g < L else 0
                b = letters[code]
                if b != "N" and het_every and (g + 1) % het_every == 0 and (i & 1):
                    b = "ACGT"[(code) % 4]  # the alternative allele of a heterozygous site, on every other template
                nxt = letters[int(ref_codes[g + 1])] if g + 1 < L else "N"
                prv = letters[int(ref_codes[g - 1])] if g > 0 else "N"
                if c2t and b == "C":
                    if rng.random() < (1 - meth_cpg if nxt == "G" else conv):
                        b = "T"
                elif not c2t and b == "G":
                    if rng.random() < (1 - meth_cpg if prv == "C" else conv):
                        b = "A"
                if b != "N" and rng.random() < err:
                    b = "ACGT"[int(rng.integers(0, 4))]
                seq.append(b)
            mates.append((p0, "".join(seq), [int(q) for q in rng.integers(20, 44, read_len)]))
        name = "t%07d" % i
        tag = aux_char(strand_tag, "C" if c2t else "G")
        first_is_r1 = fwd_first
        for k, (p0, seq, qual) in enumerate(mates):
            rev = k == 1
            r1 = first_is_r1 if k == 0 else not first_is_r1
            flag = 1 | 2 | (16 if rev else 32) | (64 if r1 else 128)
            other = mates[1 - k][0]
            recs.append(dict(name=name, flag=flag, tid=tid, pos=p0, mapq=60, cigar=[("M", read_len)], mtid=tid, mpos=other,
                             tlen=(insert if not rev else -insert), seq=seq, qual=qual, aux=tag))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    return recs
