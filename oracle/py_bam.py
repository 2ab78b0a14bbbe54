"""TEST INFRASTRUCTURE — independent pure-Python restatement of the reference's reader, the checker of csrc/bamio.c:
  parse_bam()               the BAM layout (SAM specification 4.1-4.2), through Python's gzip (a BGZF file is a multi-member gzip)
  get_next_align_details()  src/input_sam.c:222-312 with get_bam_misms (:90-136), get_seq_and_qual (:61-88), get_bs_strand (:144-220)
  read_input()              src/get_template_vector.c:49-389 as a generator of blocks
Templates are the dicts of oracle/py_prep.py: pos [fwd, rev], span [2], reads [list | None] * 2, misms [[type, position, size]] * 2,
mapq [2], orientation, bs_strand.  htslib itself is not available here: the BAM layout is the specification's."""
import gzip
import struct

PAIRED, PROPER, UNMAP, MUNMAP, REVERSE, READ2, SECONDARY, QCFAIL, DUP, SUPP = 1, 2, 4, 8, 16, 128, 256, 512, 1024, 2048
(FLT_NONE, FLT_UNMAPPED, FLT_QC, FLT_SECONDARY, FLT_MATE_UNMAPPED, FLT_DUPLICATE, FLT_NOPOS, FLT_NOMATEPOS, FLT_MISMATCH_CHR, FLT_ORIENTATION,
 FLT_INSERT_SIZE, FLT_NOSEQ, FLT_MAPQ, FLT_NOT_ALIGNED, FLT_PAIR_NOT_FOUND) = range(15)
MISMS, INS, DEL, SOFT = 0, 1, 2, 3
MAX_QUAL = 43


def parse_bam(path):
    raw = gzip.open(path, "rb").read()
    assert raw[:4] == b"BAM\1"
    l_text = struct.unpack_from("<I", raw, 4)[0]
    text = raw[8 : 8 + l_text].decode()
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, o)[0]
    o += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<I", raw, o)[0]
        name = raw[o + 4 : o + 4 + ln - 1].decode()
        refs.append((name, struct.unpack_from("<I", raw, o + 4 + ln)[0]))
        o += 8 + ln
    recs = []
    while o < len(raw):
        bs = struct.unpack_from("<I", raw, o)[0]
        tid, pos, l_name, mapq, _bin, n_cig, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHIiii", raw, o + 4)
        p = o + 36
        name = raw[p : p + l_name]
        p += l_name
        cigar = [(c & 15, c >> 4) for c in struct.unpack_from("<%dI" % n_cig, raw, p)]
        p += 4 * n_cig
        seq4 = raw[p : p + (l_seq + 1) // 2]
        p += (l_seq + 1) // 2
        qual = raw[p : p + l_seq]
        p += l_seq
        aux = raw[p : o + 4 + bs]
        recs.append(dict(tid=tid, pos=pos, name=name, mapq=mapq, cigar=cigar, flag=flag, l_seq=l_seq, mtid=mtid, mpos=mpos, tlen=tlen, seq4=seq4,
                         qual=qual, aux=aux))
        o += 4 + bs
    return text, refs, recs


def parse_sam(path):
    """SAM text (plain or gzip / BGZF) -> (header text, refs, records) in the shape parse_bam gives (SAM specification 1.3-1.5)."""
    raw = open(path, "rb").read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    text, refs, recs = [], [], []
    codes = "=ACMGRSVTWYHKDBN"
    for ln in raw.decode().split("\n"):
        ln = ln.rstrip("\r")
        if not ln:
            continue
        if ln[0] == "@" and not recs:
            text.append(ln + "\n")
            if ln.startswith("@SQ\t"):
                kv = dict(f.split(":", 1) for f in ln.split("\t")[1:] if ":" in f)
                refs.append((kv["SN"], int(kv["LN"])))
            continue
        f = ln.split("\t")
        names = [r[0] for r in refs]
        tid = -1 if f[2] == "*" else names.index(f[2])
        mtid = tid if f[6] == "=" else (-1 if f[6] == "*" else names.index(f[6]))
        cigar = []
        if f[5] != "*":
            num = ""
            for ch in f[5]:
                if ch.isdigit():
                    num += ch
                else:
                    cigar.append(("MIDNSHP=X".index(ch), int(num)))
                    num = ""
        seq = "" if f[9] == "*" else f[9].upper()
        seq4 = bytearray((len(seq) + 1) // 2)
        for i, ch in enumerate(seq):
            seq4[i >> 1] |= (codes.index(ch) if ch in codes else 15) << (4 if i % 2 == 0 else 0)
        qual = bytes([0xFF] * len(seq)) if f[10] == "*" else bytes(ord(c) - 33 for c in f[10])
        aux = b""
        for t in f[11:]:
            tag, ty, v = t[:2], t[3], t[5:]
            if ty == "A":
                aux += tag.encode() + b"A" + v[0].encode()
            elif ty == "i":
                aux += tag.encode() + b"i" + struct.pack("<i", int(v))
            elif ty in "ZH":
                aux += tag.encode() + ty.encode() + v.encode() + b"\0"
            elif ty == "f":
                aux += tag.encode() + b"f" + struct.pack("<f", float(v))
        recs.append(dict(tid=tid, pos=int(f[3]) - 1, name=f[0].encode() + b"\0", mapq=int(f[4]), cigar=cigar, flag=int(f[1]), l_seq=len(seq), mtid=mtid,
                         mpos=int(f[7]) - 1, tlen=int(f[8]), seq4=bytes(seq4), qual=qual, aux=aux))
    return "".join(text), refs, recs


def get_bs_strand(aux):
    strand, s, end = 0, 0, len(aux)
    sizes = {ord("A"): 1, ord("C"): 1, ord("c"): 1, ord("s"): 2, ord("S"): 2, ord("i"): 4, ord("I"): 4, ord("f"): 4, ord("d"): 8,
             ord("Z"): ord("Z"), ord("H"): ord("H"), ord("B"): ord("B")}
    ok = True
    while ok and s + 4 <= end:
        t0, t1 = chr(aux[s]), chr(aux[s + 1])
        al = {"ZB": "NOVALIGN", "ZS": "BSMAP", "XG": "BOWTIE", "XB": "GEM", "YD": "BWAMETH"}.get(t0 + t1)
        s += 2
        typ = chr(aux[s])
        s += 1
        if typ == "A":
            if al == "GEM":
                strand = {"C": 1, "G": 2}.get(chr(aux[s]), strand)
            s += 1
        elif typ in "Cc":
            s += 1
        elif typ in "Ss":
            if s + 2 <= end:
                s += 2
            else:
                ok = False
        elif typ in "Iif":
            if s + 4 <= end:
                s += 4
            else:
                ok = False
        elif typ == "d":
            if s + 8 <= end:
                s += 8
            else:
                ok = False
        elif typ in "ZH":
            if typ == "Z" and al:
                c = chr(aux[s]) if s < end else ""
                if al in ("BOWTIE", "NOVALIGN"):
                    strand = {"C": 1, "G": 2}.get(c, strand)
                elif al == "BSMAP":
                    strand = {"+": 1, "-": 2}.get(c, strand)
                elif al == "BWAMETH":
                    strand = {"f": 1, "r": 2}.get(c, strand)
            while s < end and aux[s]:
                s += 1
            if s < end:
                s += 1
            else:
                ok = False
        elif typ == "B":
            sz = sizes.get(aux[s], 0)
            s += 1
            if s + 4 <= end and sz:
                n = struct.unpack_from("<I", aux, s)[0]
                s += 4
                if s + n * sz <= end:
                    s += n * sz
                else:
                    ok = False
            else:
                ok = False
    return strand


def get_next_align_details(b, mapq_thresh, max_template_len, keep_unmatched, ignore_dup):
    """-> (ret, al, reverse, filtered, align_length, alignment_flag); al holds ONE read (index 1 for a reverse-strand record)."""
    flag = b["flag"]
    filtered = FLT_NONE
    if (flag & PAIRED) and not keep_unmatched:
        if (flag & (PROPER | UNMAP | MUNMAP | QCFAIL | SECONDARY | SUPP | DUP)) != PROPER:
            if flag & (SECONDARY | SUPP):
                filtered = FLT_SECONDARY
            elif flag & UNMAP:
                filtered = FLT_UNMAPPED
            elif flag & MUNMAP:
                filtered = FLT_MATE_UNMAPPED
            elif flag & QCFAIL:
                filtered = FLT_QC
            elif flag & DUP:
                if not ignore_dup:
                    filtered = FLT_DUPLICATE
            else:
                filtered = FLT_NOT_ALIGNED
    elif flag & (UNMAP | QCFAIL | SECONDARY | SUPP | DUP):
        if flag & (SECONDARY | SUPP):
            filtered = FLT_SECONDARY
        elif flag & UNMAP:
            filtered = FLT_UNMAPPED
        elif flag & QCFAIL:
            filtered = FLT_QC
        elif flag & DUP:
            filtered = FLT_DUPLICATE
    mis_matched = (flag & (MUNMAP | PROPER)) != PROPER
    reverse = bool(flag & REVERSE)
    second = bool(flag & READ2)
    al = {"pos": [0, 0], "span": [0, 0], "reads": [None, None], "misms": [[], []], "mapq": [0, 0], "bs_strand": 0,
          "orientation": 0 if ((second and reverse) or not (second or reverse)) else 1}
    mult_seg = (flag & (PAIRED | MUNMAP)) == PAIRED
    u32 = 0xFFFFFFFF  # forward_position / reverse_position are uint32_t: a negative mate position wraps (include/bs_call.h:66-67)
    if reverse:
        al["pos"] = [(b["mpos"] + 1) & u32, (b["pos"] + 1) & u32]
        al["mapq"][1] = b["mapq"]
    else:
        al["pos"] = [(b["pos"] + 1) & u32, (b["mpos"] + 1) & u32]
        al["mapq"][0] = b["mapq"]
    if b["mapq"] < mapq_thresh and not filtered:
        filtered = FLT_MAPQ
    aflag = flag
    if mult_seg:
        if b["tid"] != b["mtid"]:
            if not filtered:
                filtered = FLT_MISMATCH_CHR
            if keep_unmatched:
                mis_matched = True
        if not filtered and abs(b["tlen"]) > max_template_len:
            filtered = FLT_INSERT_SIZE
            if keep_unmatched:
                mis_matched = True
        if reverse:
            if b["pos"] < b["mpos"]:
                if not filtered:
                    filtered = FLT_ORIENTATION
                if keep_unmatched:
                    mis_matched = True
            if mis_matched:
                al["pos"][0] = 0
        else:
            if b["pos"] > b["mpos"]:
                if not filtered:
                    filtered = FLT_ORIENTATION
                if keep_unmatched:
                    mis_matched = True
            if mis_matched:
                al["pos"][1] = 0
    if not mult_seg or mis_matched:
        aflag &= ~PAIRED
    ret = 0
    if filtered and not (keep_unmatched and filtered in (FLT_INSERT_SIZE, FLT_MISMATCH_CHR, FLT_ORIENTATION)):
        ret = 1
    align_length = 0
    if ret == 0:
        ix = 1 if reverse else 0
        span = position = 0
        ms = []
        for op, ln in b["cigar"]:  # MIDNSHP=X
            if op in (0, 7, 8):
                position += ln
                span += ln
            elif op in (6, 4):
                ms.append([SOFT, position, ln])
                position += ln
            elif op == 1:
                ms.append([DEL, position, ln])
                position += ln
            elif op == 2:
                ms.append([INS, position, ln])
                span += ln
        al["misms"][ix] = ms
        al["span"][ix] = span
        align_length = position
        rd = []
        for k in range(b["l_seq"]):
            c4 = (b["seq4"][k >> 1] >> (0 if k & 1 else 4)) & 15
            code = {1: 1, 2: 2, 4: 3, 8: 4}.get(c4, 0)
            q = min(b["qual"][k], MAX_QUAL)
            rd.append(((code - 1) | (q << 2)) if code else 0)
        al["reads"][ix] = rd
        al["bs_strand"] = get_bs_strand(b["aux"])
    return ret, al, reverse, filtered, align_length, aflag


def get_al_qual(al):
    from . import py_prep

    return py_prep.get_al_qual(al)


def region_query(recs, tid, start, stop):
    """What sam_itr_queryi(idx, tid, start - 1, stop) iterates over (src/get_template_vector.c:69-74): the records of `tid` that
    overlap [start - 1, stop) — an alignment without reference-consuming operations counts as one base long."""
    out = []
    for b in recs:
        reflen = sum(ln for op, ln in b["cigar"] if op in (0, 2, 3, 7, 8))
        if b["tid"] == tid and b["pos"] < stop and b["pos"] + (reflen or 1) > start - 1:
            out.append(b)
    return out


def read_input(recs, mapq_thresh=20, max_template_len=1000, keep_unmatched=False, ignore_duplicates=False, keep_duplicates=False,
               stats=None):
    """Generator of (tid, y, [templates]) — the blocks read_input queues for the process thread."""
    if stats is None:
        stats = {"cts": [0] * 15, "bases": [0] * 15}
    curr_tid = old_tid = -1
    max_pos = start_pos = 0
    read_idx = curr_pos = start_idx = 0
    align_list, al_hash_list = [], []
    hash_base = {}  # tag -> {"al", "flag", "ix"}
    for b in recs:
        ret, al, reverse, filtered, align_length, aflag = get_next_align_details(b, mapq_thresh, max_template_len, keep_unmatched, ignore_duplicates)
        if ret > 0:
            stats["cts"][filtered] += 1
            stats["bases"][filtered] += b["l_seq"]
            continue
        tag = b["name"]
        new_block = new_contig = False
        if curr_tid < 0 or curr_tid != b["tid"]:
            new_contig = new_block = True
            old_tid, curr_tid = curr_tid, b["tid"]
        insert = True
        fwd, rev = al["pos"]
        if not new_contig:
            if (aflag & PAIRED) and fwd > 0 and rev > 0:
                if fwd == rev:
                    insert = tag not in hash_base
                elif reverse:
                    insert = fwd > rev
                else:
                    insert = fwd < rev
            if insert and start_pos > 0:
                if fwd > 0:
                    if fwd > max_pos and (rev > max_pos or rev == 0):
                        if fwd - max_pos > 1:
                            new_block = True
                elif rev > max_pos and rev - max_pos > 1:
                    new_block = True
        if new_block:
            hash_base.clear()
            assert insert
            read_idx = start_idx = curr_pos = 0
            if align_list:
                yield (old_tid if new_contig else curr_tid), max_pos, align_list
                align_list, al_hash_list = [], []
            if new_contig and old_tid >= 0:
                old_tid = -1
            max_pos = start_pos = 0
        ix = 1 if reverse else 0
        st = al["pos"][ix]
        ml = st + al["span"][ix]
        max_pos = max(max_pos, ml)
        if start_pos == 0 or start_pos > st:
            start_pos = st

        def append(al_, h):
            nonlocal read_idx
            align_list.append(al_)
            al_hash_list.append(h)
            read_idx += 1

        if aflag & PAIRED:
            if not insert:
                th = hash_base.pop(tag, None)
                if th is not None:
                    t = th["al"]
                    t["reads"][ix] = al["reads"][ix]
                    t["mapq"][ix] = al["mapq"][ix]
                    t["span"][ix] = al["span"][ix]
                    t["misms"][ix] = al["misms"][ix]
                    assert t["pos"] == al["pos"]
                    al_hash_list[th["ix"]] = None
                else:
                    stats["cts"][14] += 1
                    stats["bases"][14] += len(al["reads"][ix])
                    al_skip = False
                    if not keep_duplicates:
                        if (rev if reverse else fwd) >= start_pos:
                            al_skip = True
                    if not al_skip and keep_unmatched:
                        x = (fwd if fwd > 0 else rev) + align_length
                        max_pos = max(max_pos, x)
                        append(al, None)
            else:
                al_skip = False
                if not keep_duplicates:
                    pos = fwd if fwd > 0 else rev
                    if pos == curr_pos:
                        for j in range(start_idx, read_idx):
                            al1 = align_list[j]
                            if al["pos"] == al1["pos"] and al["bs_strand"] == al1["bs_strand"]:
                                def meanq(a):
                                    qs = [a["mapq"][z] for z in range(2) if a["reads"][z]]
                                    return sum(qs) // len(qs)

                                maxq, maxq1 = meanq(al), meanq(al1)
                                if maxq1 < maxq or (maxq == maxq1 and get_al_qual(al1) < get_al_qual(al)):
                                    th = hash_base.get(tag)
                                    assert not (th is not None and al_hash_list[j] is not None)
                                    if th is None:
                                        th = al_hash_list[j]
                                    align_list[j] = al
                                    if th is not None:
                                        for key in [k_ for k_, v in hash_base.items() if v is th]:
                                            del hash_base[key]
                                        th.update(al=al, flag=aflag, ix=j)
                                    else:
                                        th = {"al": al, "flag": aflag, "ix": j}
                                    hash_base[tag] = th
                                    al = al1
                                l1 = len(al["reads"][0] or [])
                                l2 = len(al["reads"][1] or [])
                                stats["cts"][FLT_DUPLICATE] += 2 if (l1 and l2) else 1
                                stats["bases"][FLT_DUPLICATE] += l1 + l2
                                al_skip = True
                    else:
                        curr_pos = pos
                        start_idx = read_idx
                if not al_skip:
                    assert tag not in hash_base
                    th = {"al": al, "flag": aflag, "ix": read_idx}
                    hash_base[tag] = th
                    append(al, th)
        else:
            al_skip = False
            if not keep_duplicates:
                pos = fwd if fwd > 0 else rev
                if pos == curr_pos:
                    for j in range(start_idx, read_idx):
                        al1 = align_list[j]
                        th = al_hash_list[j]
                        if al["pos"] == al1["pos"] and al["bs_strand"] == al1["bs_strand"] and (th is None or (th["flag"] & 9) in (9, 0)):
                            if al1["mapq"][0] < al["mapq"][0] or (al1["mapq"][0] == al["mapq"][0] and get_al_qual(al1) < get_al_qual(al)):
                                align_list[j] = al
                                al = al1
                            stats["cts"][FLT_DUPLICATE] += 1
                            stats["bases"][FLT_NONE] += len(al["reads"][ix] or [])
                            al_skip = True
                else:
                    curr_pos = pos
                    start_idx = read_idx
            if not al_skip:
                append(al, None)
    if align_list:
        yield curr_tid, max_pos, align_list


def check_block(als, y):
    """What process_template_vector asserts before it touches a block (src/process_template.c:22-26)."""
    x = als[0]["pos"][0] or als[0]["pos"][1]
    assert x > 0 and x <= y
