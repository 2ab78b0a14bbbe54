#!/usr/bin/env python3
"""Corrupted-input fuzz of the library's HOST readers (csrc/bamio.c: BAM / SAM text / FASTA; csrc/bamstream.c + csrc/inflate_fast.c: the
device reader's host half over the same damaged BAM files; csrc/dbsnp.c: the dbSNP index;
csrc/prep.c behind the reader): valid files written by tools/make_bam.py / tools/make_dbsnp_index.py are damaged — bytes of
the UNCOMPRESSED payload flipped, 32-bit fields set to extreme values, pieces cut out or repeated, then compressed again so
the damage reaches the parsers and not only the inflate / CRC checks; and the same damage done to the compressed files — and
every reader must either finish or fail with a BscError.  Anything else (a crash, a Python error out of the wrappers, a
sanitizer report when run under the sanitized build, tests/test_host_sanitizers.py) is a finding.  CPU only: no GPU call.
usage: python tools/fuzz_host_inputs.py [--seed S] [--rounds N] [--seconds T] [--dir D]"""
import argparse
import importlib.util
import os
import struct
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bs_call_amd.bam import BamReader, fasta_contig  # noqa: E402
from bs_call_amd.caller import BscError, prepare_templates  # noqa: E402
from bs_call_amd.dbsnp import DbSnpIndex  # noqa: E402


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


W = _load("make_bam")
D = _load("make_dbsnp_index")
REFS = [("chr1", 100_000), ("chr2", 50_000)]
EXTREME = (0, 1, 0x7F, 0x80, 0xFF, 0x7FFF, 0x8000, 0xFFFF, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0xFFFFFFFE, 0x10000000)


def damage(rng, data: bytes) -> bytes:
    """1-4 random edits of a byte string"""
    b = bytearray(data)
    for _ in range(int(rng.integers(1, 5))):
        if not b:
            break
        kind = int(rng.integers(0, 7))
        o = int(rng.integers(0, len(b)))
        if kind == 0:  # flip a byte
            b[o] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:  # a random byte
            b[o] = int(rng.integers(0, 256))
        elif kind == 2 and len(b) >= 4:  # an extreme 32-bit value somewhere (lengths, positions, counts)
            o = min(o, len(b) - 4)
            b[o : o + 4] = struct.pack("<I", EXTREME[int(rng.integers(0, len(EXTREME)))])
        elif kind == 3 and len(b) >= 2:  # an extreme 16-bit value
            o = min(o, len(b) - 2)
            b[o : o + 2] = struct.pack("<H", EXTREME[int(rng.integers(0, 8))])
        elif kind == 4:  # cut a piece out
            n = int(rng.integers(1, 64))
            del b[o : o + n]
        elif kind == 5:  # repeat a piece
            n = int(rng.integers(1, 64))
            b[o:o] = b[o : o + n]
        else:  # truncate
            del b[o:]
    return bytes(b)


def random_records(rng, n):
    recs, pos = [], 50
    for i in range(n):
        pos += int(rng.integers(0, 40))
        ln = int(rng.integers(20, 120))
        seq = "".join("ACGTN"[int(c)] for c in rng.integers(0, 5, ln))
        paired = rng.random() < 0.7
        cig = [("M", ln)]
        if rng.random() < 0.3 and ln > 30:
            a = int(rng.integers(1, 10))
            cig = [("S", a), ("M", ln - a - 6), ("I", 2), ("M", 4)] if rng.random() < 0.5 else [("M", 10), ("D", 3), ("M", ln - 10)]
        aux = W.aux_char("XB", "CG"[int(rng.integers(0, 2))]) + W.aux_int("NM", 1)
        if paired:
            gap = int(rng.integers(0, 300))
            recs.append(dict(name="p%d" % i, flag=99, tid=0, pos=pos, mapq=int(rng.integers(0, 61)), cigar=cig, mtid=0, mpos=pos + gap, tlen=gap + ln,
                             seq=seq, qual=[int(q) for q in rng.integers(2, 42, ln)], aux=aux))
            recs.append(dict(name="p%d" % i, flag=147, tid=0, pos=pos + gap, mapq=int(rng.integers(0, 61)), cigar=[("M", ln)], mtid=0, mpos=pos, tlen=-(gap + ln),
                             seq=seq[::-1], qual=[int(q) for q in rng.integers(2, 42, ln)], aux=aux))
        else:
            recs.append(dict(name="s%d" % i, flag=0 if rng.random() < 0.5 else 16, tid=0, pos=pos, mapq=60, cigar=cig, mtid=-1, mpos=-1, tlen=0, seq=seq,
                             qual=[int(q) for q in rng.integers(2, 42, ln)], aux=aux))
    recs.sort(key=lambda r: r["pos"])
    return recs


def bam_payload(refs, records):
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    t = text.encode()
    data = bytearray(b"BAM\1" + struct.pack("<I", len(t)) + t + struct.pack("<i", len(refs)))
    for name, ln in refs:
        data += struct.pack("<I", len(name) + 1) + name.encode() + b"\0" + struct.pack("<I", ln)
    head = len(data)
    for r in records:
        data += W.encode_record(r)
    return bytes(data), head


def write_bgzf(path, data, block):
    with open(path, "wb") as f:
        for o in range(0, len(data), block):
            f.write(W.bgzf_block(data[o : o + block]))
        f.write(W.BGZF_EOF)


def drain_reader(path, prep=None, **kw):
    """open, read every block, touch every array; BscError = a clean refusal"""
    try:
        with BamReader(path, **kw) as r:
            _ = r.refs, r.header_text
            n = 0
            for tid, y, tpl, seq, ms in r.blocks():
                n += len(tpl) + int(seq.sum() & 1) + len(ms)
                if prep is not None:  # the process thread's step on what the reader delivered, and on a damaged copy of it
                    prep(tpl, seq, ms)
            r.filter_counts()
        return "ok"
    except BscError:
        return "refused"


def drain_stream(path, rng):
    """the device reader's host half (csrc/bamstream.c + csrc/inflate_fast.c: the BGZF block index, the helpers' own inflate and CRC-32, the
    speculative record walk and its check) over the same damaged file, whole or by contig: every slab touched; BscError = a clean refusal"""
    from bs_call_amd.bamdev import BamStream

    try:
        kw = {}
        if rng.random() < 0.3:
            kw["contigs"] = [int(v) for v in rng.choice([-1, 0, 1, 2, 7], size=int(rng.integers(0, 3)), replace=False)]
        with BamStream(path, threads=int(rng.integers(1, 4)), slab_bytes=int(rng.choice([0, 65536, 1 << 20])), n_slabs=int(rng.choice([0, 2, 3])), **kw) as s:
            n = 0
            for off, data, offs, last in s.slabs():
                n += len(data) + int(offs.sum() & 1)
            _ = s.refs, s.header_text, s.first_record
        return "ok"
    except BscError:
        return "refused"


def prep_step(rng, stats):
    """bsc_prepare_templates (csrc/prep.c) on a block as read, then with the block's arrays damaged: offsets, lengths, mismatch
    lists and positions are the caller's, and the entry must refuse what does not fit its buffers"""

    def run(tpl, seq, ms):
        lt, rt = (int(rng.integers(0, 12)), int(rng.integers(0, 12))), (int(rng.integers(0, 12)), int(rng.integers(0, 12)))
        out = prepare_templates(tpl, seq, ms, lt, rt, int(rng.integers(1, 40)))
        assert len(out[0]) == len(tpl)
        stats["prep_ok"] += 1
        for _ in range(3):
            t2, s2, m2 = tpl.copy(), seq.copy(), ms.copy()
            which = int(rng.integers(0, 3))
            arr = (t2, s2, m2)[which]
            if arr.size == 0:
                continue
            raw = np.frombuffer(damage_fixed(rng, arr.tobytes()), dtype=arr.dtype).copy()
            if which == 0:
                t2 = raw
            elif which == 1:
                s2 = raw
            else:
                m2 = raw
                # a listed gap is padded base by base, as the reference does (src/al_utils.c:164-215): 4 G of it is 4 GB of
                # honest work, not a finding — keep the damaged sizes where a run of this tool stays small
                m2["size"] = np.minimum(m2["size"], 1 << 20)
            try:
                prepare_templates(t2, s2, m2, lt, rt, 20)
                stats["prep_damaged_ok"] += 1
            except BscError:
                stats["prep_damaged_refused"] += 1

    return run


def damage_fixed(rng, data: bytes) -> bytes:
    """edits that keep the length (arrays of fixed-size records)"""
    b = bytearray(data)
    for _ in range(int(rng.integers(1, 4))):
        o = int(rng.integers(0, len(b)))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            b[o] = int(rng.integers(0, 256))
        elif kind == 1 and len(b) >= 4:
            o = min(o, len(b) - 4) & ~3
            b[o : o + 4] = struct.pack("<I", EXTREME[int(rng.integers(0, len(EXTREME)))])
        elif len(b) >= 8:
            o = min(o, len(b) - 8) & ~7
            b[o : o + 8] = struct.pack("<Q", (0, 1, 0xFFFFFFFFFFFFFFFF, 0x7FFFFFFFFFFFFFFF, 0x8000000000000000, 1 << 32, (1 << 32) - 1)[int(rng.integers(0, 7))])
    return bytes(b)


def one_bam(rng, d, stats):
    recs = random_records(rng, int(rng.integers(5, 120)))
    data, head = bam_payload(REFS, recs)
    p = os.path.join(d, "f.bam")
    mode = int(rng.integers(0, 5))
    if mode == 4:  # an intact file: the damage is done to the arrays between the reader and the pre-processing (prep_step)
        write_bgzf(p, data, int(rng.integers(200, 0xFF00)))
    elif mode == 0:  # damage inside the records
        body = damage(rng, data[head:])
        write_bgzf(p, data[:head] + body, int(rng.integers(200, 0xFF00)))
    elif mode == 1:  # damage inside the header
        write_bgzf(p, damage(rng, data[:head]) + data[head:], 0xFF00)
    elif mode == 2:  # damage of the compressed file
        write_bgzf(p, data, int(rng.integers(200, 0xFF00)))
        raw = open(p, "rb").read()
        open(p, "wb").write(damage(rng, raw))
    else:  # a record cut by a block boundary at every possible place: small blocks
        write_bgzf(p, damage(rng, data), int(rng.integers(30, 400)))
    stats["bam_" + drain_reader(p, prep=prep_step(rng, stats), threads=int(rng.integers(0, 3)))] += 1
    stats["bamstream_" + drain_stream(p, rng)] += 1


def one_sam(rng, d, stats):
    recs = random_records(rng, int(rng.integers(5, 60)))
    p = os.path.join(d, "f.sam")
    W.write_sam(p, REFS, recs)
    raw = open(p, "rb").read()
    mode = int(rng.integers(0, 3))
    if mode == 0:
        raw = damage(rng, raw)
    elif mode == 1:  # a field replaced by something a line parser may choke on
        lines = raw.split(b"\n")
        i = int(rng.integers(0, len(lines)))
        f = lines[i].split(b"\t")
        j = int(rng.integers(0, len(f)))
        f[j] = [b"", b"*", b"-1", b"99999999999999999999", b"4294967295", b"2147483648", b"0", b"1000000M", b"5M5", b"M", b"XX:i:", b"XX:Z", b"XX:B:i,", b"\xff\xfe",
                b"A" * 70000][int(rng.integers(0, 15))]
        lines[i] = b"\t".join(f)
        raw = b"\n".join(lines)
    else:  # fields dropped from the end of a line
        lines = raw.split(b"\n")
        i = int(rng.integers(0, len(lines)))
        f = lines[i].split(b"\t")
        lines[i] = b"\t".join(f[: int(rng.integers(0, len(f) + 1))])
        raw = b"\n".join(lines)
    if rng.random() < 0.3:
        write_bgzf(p, raw, int(rng.integers(100, 5000)))
    else:
        open(p, "wb").write(raw)
    stats["sam_" + drain_reader(p)] += 1


def one_fasta(rng, d, stats):
    p = os.path.join(d, "f.fa")
    text = b">chr1 x\n" + b"\n".join(bytes(rng.choice(list(b"ACGTNacgtn"), 60).tolist()) for _ in range(int(rng.integers(1, 30)))) + b"\n>chr2\nACGT\n"
    raw = damage(rng, text)
    if rng.random() < 0.3:
        import gzip

        raw = gzip.compress(raw)
        if rng.random() < 0.5:
            raw = damage(rng, raw)
    open(p, "wb").write(raw)
    for name in ("chr1", "chr2"):
        try:
            fasta_contig(p, name, length_hint=int(rng.integers(0, 100)))
            stats["fasta_ok"] += 1
        except BscError:
            stats["fasta_refused"] += 1


def one_dbsnp(rng, d, stats):
    p = os.path.join(d, "f.idx")
    sites = {"chr1": D.synthetic_sites(int(rng.integers(2000, 60000)), spacing=int(rng.integers(20, 400)), seed=int(rng.integers(1, 1 << 40))),
             "chr2": D.synthetic_sites(5000, spacing=100, seed=int(rng.integers(1, 1 << 40)))}
    mode = int(rng.integers(0, 3))
    orig = D.zlib.compress
    if mode == 0:  # damage inside the uncompressed blocks / the directory
        which = int(rng.integers(0, 4))
        calls = [0]

        def comp(b, *a):
            calls[0] += 1
            return orig(damage(rng, b) if (calls[0] - 1) % 4 == which else b, *a)

        D.zlib.compress = comp
    try:
        D.write_index(p, sites)
    finally:
        D.zlib.compress = orig
    if mode == 1:  # the compressed file
        open(p, "wb").write(damage(rng, open(p, "rb").read()))
    elif mode == 2:  # the fixed header (magic, offsets, sizes)
        raw = bytearray(open(p, "rb").read())
        o = int(rng.integers(0, 8)) * 4
        raw[o : o + 4] = struct.pack("<I", EXTREME[int(rng.integers(0, len(EXTREME)))])
        open(p, "wb").write(bytes(raw))
    try:
        with DbSnpIndex(p) as ix:
            _ = ix.contigs, ix.header
            for name in ("chr1", "chr2", "chrX"):
                n = ix.load_contig(name)
                if n:
                    ix.flags(1, 70000).sum()
                    for x in (1, 63, 64, 1000, 59999):
                        ix.name(x)
                    pos, off, by = ix.names(1, 70000)  # the table the device BCF encoder searches (bsc_dbsnp_names)
                    assert len(off) == len(pos) + 1 and int(off[-1]) == len(by) and (np.diff(pos.astype(np.int64)) > 0).all()
        stats["dbsnp_ok"] += 1
    except BscError:
        stats["dbsnp_refused"] += 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=400)
    ap.add_argument("--seconds", type=float, default=0.0, help="stop after this long instead of after --rounds")
    ap.add_argument("--dir", default=None)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    import collections

    stats = collections.Counter()
    t0 = time.time()
    with tempfile.TemporaryDirectory(dir=a.dir) as d:
        i, slow = 0, (0.0, -1, "")
        while (time.time() - t0 < a.seconds) if a.seconds else (i < a.rounds):
            t1 = time.time()
            (one_bam, one_sam, one_fasta, one_dbsnp)[i % 4](rng, d, stats)
            if time.time() - t1 > slow[0]:
                slow = (time.time() - t1, i, ("bam", "sam", "fasta", "dbsnp")[i % 4])
            i += 1
    print("fuzz_host_inputs: seed %d, %d damaged files in %.1f s, every reader finished or refused: %s; slowest file: %.2f s (round %d, %s)"
          % (a.seed, i, time.time() - t0, dict(sorted(stats.items())), slow[0], slow[1], slow[2]))


if __name__ == "__main__":
    main()
