"""csrc/bamio.c (BAM in, blocks of templates out) against hand-worked scenarios — one per branch of get_next_align_details
and read_input, each derived from the cited lines — and against the independent Python restatement (oracle/py_bam.py) on
random coordinate-sorted BAM files written by tools/make_bam.py.  Host code: CPU only."""
import importlib.util
import os

import numpy as np
import pytest

from bs_call_amd.bam import BamReader
from bs_call_amd.caller import BscError
from oracle import py_bam

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("make_bam", os.path.join(ROOT, "tools", "make_bam.py"))
W = importlib.util.module_from_spec(spec)
spec.loader.exec_module(W)

REFS = [("chr1", 100_000), ("chr2", 50_000)]


def rec(name, flag, pos, mpos, seq="ACGTACGTAC", cigar=None, mapq=60, tid=0, mtid=None, tlen=0, qual=None, aux=b"", **kw):
    cigar = cigar or [("M", len(seq))]
    return dict(name=name, flag=flag, tid=tid, pos=pos, mapq=mapq, cigar=cigar, mtid=tid if mtid is None else mtid, mpos=mpos, tlen=tlen,
                seq=seq, qual=None if qual == "missing" else (qual if qual is not None else [30] * len(seq)), aux=aux, **kw)


THREADS = [0]  # test_threaded_reader re-runs the comparisons with helper threads


def c_blocks(path, region=None, **kw):
    out = []
    with BamReader(path, threads=THREADS[0], region=region, **kw) as r:
        for tid, y, tpl, seq, ms in r.blocks():
            ts = []
            for t in tpl:
                reads, misms = [], []
                for k in range(2):
                    ln = int(t["len"][k])
                    reads.append(seq[int(t["off"][k]) : int(t["off"][k]) + ln].tolist() if ln else None)
                    o, n = int(t["misms_off"][k]), int(t["n_misms"][k])
                    misms.append([[int(m["type"]), int(m["position"]), int(m["size"])] for m in ms[o : o + n]])
                ts.append({"pos": [int(v) for v in t["pos"]], "span": [int(t["reference_span"][k]) if t["len"][k] else 0 for k in range(2)],
                           "reads": reads, "misms": misms, "mapq": [int(v) for v in t["mapq"]], "orientation": int(t["orientation"]),
                           "bs_strand": int(t["bs_strand"])})
            out.append((tid, y, ts))
        cts, bases = r.filter_counts()
    return out, cts, bases


def py_blocks(path, region=None, **kw):
    text, refs, recs = py_bam.parse_bam(path)
    if region:
        recs = py_bam.region_query(recs, *region)
    st = {"cts": [0] * 15, "bases": [0] * 15}
    out = []
    for tid, y, als in py_bam.read_input(recs, stats=st, **kw):
        py_bam.check_block(als, y)  # the library's reader reports what the process thread would assert on
        out.append((tid, y, [{"pos": list(a["pos"]), "span": [a["span"][k] if a["reads"][k] else 0 for k in range(2)], "reads": [a["reads"][0], a["reads"][1]],
                              "misms": a["misms"], "mapq": list(a["mapq"]), "orientation": a["orientation"], "bs_strand": a["bs_strand"]} for a in als]))
    return out, st["cts"], st["bases"]


def both(tmp_path, records, **kw):
    p = str(tmp_path / "t.bam")
    W.write_bam(p, REFS, records)
    c = c_blocks(p, **kw)
    py = py_blocks(p, **kw)
    assert c == py
    return c


def b(base, q=30):
    return "ACGT".index(base) | (q << 2)


# ---- hand-worked scenarios ------------------------------------------------------------------------------------------------
def test_header_and_a_proper_pair(tmp_path):
    """Forward read 1 at 1000 (flag 99 = paired, proper, mate reverse, first), reverse read 2 at 1200 (147): one template,
    FORWARD orientation (read[0] = R1), forward_position 1001, reverse_position 1201, both MAPQs, y = 1201 + 10."""
    r1 = rec("p", 99, 1000, 1200, "ACGTACGTAC", tlen=210, aux=W.aux_char("XB", "C"), mapq=50)
    r2 = rec("p", 147, 1200, 1000, "TTTTTGGGGG", tlen=-210, aux=W.aux_char("XB", "C"), mapq=40)
    (blocks, cts, bases) = both(tmp_path, [r1, r2])
    assert blocks == [(0, 1211, [{"pos": [1001, 1201], "span": [10, 10], "reads": [[b(c) for c in "ACGTACGTAC"], [b(c) for c in "TTTTTGGGGG"]],
                                   "misms": [[], []], "mapq": [50, 40], "orientation": 0, "bs_strand": 1}])]
    assert sum(cts) == 0
    with BamReader(str(tmp_path / "t.bam")) as r:
        assert r.refs == REFS and r.header_text.startswith("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:100000\n")


def test_orientation_strand_tags_qualities_and_n(tmp_path):
    """Read 2 forward + read 1 reverse = a REVERSE template; the strand tag of each aligner (src/input_sam.c:144-220); a quality
    above 43 is clamped, an N (and any ambiguity code) is byte 0 whatever its quality; missing qualities (0xff) clamp to 43."""
    recs = [rec("a", 163, 100, 300, "ACNTR", tlen=205, qual=[50, 43, 40, 7, 30], aux=W.aux_str("ZS", "-+")),  # BSMAP '-': G2A
            rec("b", 99, 102, 302, "ACGTA", tlen=205, qual="missing", aux=W.aux_str("XG", "GA")),                 # Bowtie/Bismark: G2A
            rec("c", 0, 104, -1, "ACGTA", aux=W.aux_int("NM", 3) + W.aux_str("YD", "f")),                     # bwa-meth f: C2T, single
            rec("d", 16, 106, -1, "ACGTA", aux=W.aux_str("ZB", "CT")),                                        # Novoalign: C2T, reverse single
            rec("a", 83, 300, 100, "GGGGG", tlen=-205, aux=W.aux_str("ZS", "-+")),
            rec("b", 147, 302, 102, "CCCCC", tlen=-205, aux=W.aux_str("XG", "GA"))]
    (blocks, cts, _) = both(tmp_path, recs)
    ts = blocks[0][2]
    assert [t["orientation"] for t in ts] == [1, 0, 0, 1] and [t["bs_strand"] for t in ts] == [2, 2, 1, 1]
    assert ts[0]["reads"][0] == [b("A", 43), b("C", 43), 0, b("T", 7), 0]
    assert ts[1]["reads"][0] == [b(c, 43) for c in "ACGTA"]
    assert ts[2]["pos"] == [105, 0] and ts[2]["reads"][1] is None and ts[3]["pos"] == [0, 107] and ts[3]["reads"][0] is None
    assert ts[3]["mapq"] == [0, 60]  # a reverse single read keeps its MAPQ in slot 1


def test_cigar_to_mismatch_list(tmp_path):
    """3S 4M 2I 3M 5D 2M 1S: soft clips and the insertion are entries with their offset in the read, the deletion an INS entry
    (the reference's naming) that adds to the reference span only (src/input_sam.c:90-136)."""
    seq = "A" * 15
    r = rec("x", 0, 500, -1, seq, cigar=[("S", 3), ("M", 4), ("I", 2), ("M", 3), ("D", 5), ("M", 2), ("S", 1)])
    (blocks, _, _) = both(tmp_path, [r])
    t = blocks[0][2][0]
    assert t["misms"][0] == [[3, 0, 3], [2, 7, 2], [1, 12, 5], [3, 14, 1]] and t["span"][0] == 4 + 3 + 5 + 2
    assert blocks[0][1] == 501 + 14  # y = position + reference span


def test_flag_filters_and_their_reasons(tmp_path):
    """One record per verdict of get_next_align_details (:234-300), counted with its bases; none reaches a block."""
    ok1, ok2 = rec("ok", 99, 100, 200, tlen=110), rec("ok", 147, 200, 100, tlen=-110)
    bad = [rec("sec", 99 | 256, 101, 200), rec("unm", 1 | 4, 102, 200), rec("mun", 1 | 8, 103, 200), rec("qc", 99 | 512, 104, 200),
           rec("dup", 99 | 1024, 105, 200), rec("npp", 1 | 32 | 64, 106, 200), rec("lowq", 99, 107, 200, mapq=19),
           rec("chr", 99, 108, 200, mtid=1), rec("long", 99, 109, 2000, tlen=1901), rec("ori", 99, 210, 110, tlen=-100),
           rec("s_dup", 1024, 111, -1), rec("s_qc", 512, 112, -1)]
    recs = sorted([ok1, ok2] + bad, key=lambda r: r["pos"])
    (blocks, cts, bases) = both(tmp_path, recs)
    assert len(blocks) == 1 and len(blocks[0][2]) == 1
    exp = [0] * 15
    for reason, n in ((3, 1), (1, 1), (4, 1), (2, 2), (5, 2), (13, 1), (12, 1), (8, 1), (10, 1), (9, 1)):
        exp[reason] = n
    assert cts == exp and bases == [10 * v for v in exp]


def test_blocks_split_at_gaps_and_contigs(tmp_path):
    """A read starting more than one base beyond the rightmost covered position opens a new block (:139-147); adjacency (gap of
    one) does not; a new contig always does, and the old block keeps the old contig's id (:176-183)."""
    recs = [rec("a", 0, 100, -1), rec("b", 0, 110, -1),  # 101..110, then 111..120: touching -> same block
            rec("c", 0, 121, -1),                         # starts at 122 = max_pos(121) + 1: still the same block
            rec("d", 0, 133, -1),                         # 134 > 131 + 1: new block
            rec("e", 0, 50, -1, tid=1)]
    (blocks, _, _) = both(tmp_path, recs)
    assert [(tid, y, len(ts)) for tid, y, ts in blocks] == [(0, 132, 3), (0, 144, 1), (1, 61, 1)]


def test_duplicates_pairs_and_singles(tmp_path):
    """Templates with the same positions and strand starting at the same place: the better one stays (mean MAPQ, then
    get_al_qual), the other is counted as a duplicate — two reads if it was a complete pair (:282-321); the dropped
    template's mate then finds nobody waiting: PairNotFound (:243-246)."""
    recs = [rec("p1", 99, 100, 300, tlen=210, mapq=30), rec("p2", 99, 100, 300, tlen=210, mapq=50),  # p2 replaces p1
            rec("p3", 99, 100, 300, tlen=210, mapq=50, qual=[20] * 10),                                # same MAPQ, lower quality: dropped
            rec("s1", 0, 105, -1, mapq=40), rec("s2", 0, 105, -1, mapq=41),                            # singles: s2 replaces s1
            rec("p1", 147, 300, 100, tlen=-210, mapq=30), rec("p2", 147, 300, 100, tlen=-210, mapq=50),
            rec("p3", 147, 300, 100, tlen=-210, mapq=50)]
    (blocks, cts, bases) = both(tmp_path, recs)
    ts = blocks[0][2]
    assert len(ts) == 2 and ts[0]["mapq"] == [50, 50] and ts[0]["reads"][1] is not None and ts[1]["mapq"] == [41, 0]
    assert cts[5] == 3 and bases[5] == 20 and bases[0] == 10  # p1, p3 (one read each so far), s1 — whose bases land in the PASSED column (:363)
    assert cts[14] == 2 and bases[14] == 20                   # the mates of p1 and p3


def test_keep_duplicates_and_keep_unmatched(tmp_path):
    recs = [rec("p1", 99, 100, 300, tlen=210), rec("p2", 99, 100, 300, tlen=210), rec("far", 99, 105, 5000, tlen=4905),
            rec("p1", 147, 300, 100, tlen=-210), rec("p2", 147, 300, 100, tlen=-210)]
    (blocks, cts, _) = both(tmp_path, recs, keep_duplicates=True)
    assert len(blocks[0][2]) == 2 and cts[5] == 0 and cts[10] == 1
    (blocks, cts, _) = both(tmp_path, recs, keep_duplicates=True, keep_unmatched=True)
    ts = blocks[0][2]
    assert len(ts) == 3 and ts[2]["pos"] == [106, 0] and cts[10] == 0  # the over-long pair's read is kept as a single


def test_unsorted_and_broken_input(tmp_path):
    p = str(tmp_path / "bad.bam")
    # a backwards-facing mate whose partner never came is counted (PairNotFound) and dropped, it does not open a block
    W.write_bam(p, REFS, [rec("a", 0, 100, -1), rec("m", 147, 400, 100, tlen=-310)])
    blocks, cts, _ = c_blocks(p)
    assert [(t, y, len(ts)) for t, y, ts in blocks] == [(0, 411, 1)] and cts[14] == 1
    with open(p, "wb") as f:
        f.write(b"\x1f\x8b\x08\x00" + b"\0" * 30)
    with pytest.raises(BscError, match="BGZF"):
        BamReader(p)
    W.write_bam(p, REFS, [rec("a", 0, 100, -1)])
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[: len(raw) - 40])
    with pytest.raises(BscError):
        c_blocks(p)
    with pytest.raises(BscError, match="cannot open"):
        BamReader(str(tmp_path / "missing.bam"))


def test_counts_and_positions_of_a_damaged_file(tmp_path):
    """Found by tools/fuzz_host_inputs.py under the sanitized build (tests/test_host_sanitizers.py): (1) l_text / n_ref of a BAM
    header sized allocations before the bytes behind them were read — n_ref = 2^31 - 1 reserved 16 GB and the clean-up walked
    all of it; (2) POS / PNEXT outside 0 .. 2^31 - 1 in SAM text wrapped (signed overflow) instead of failing like sam_parse1;
    (3) a BAM record with pos = 2^31 - 1 overflowed the 1-based position."""
    import struct
    import time

    p = str(tmp_path / "h.bam")
    for l_text, n_ref in ((0x7FFFFFFF, 1), (4, 0x7FFFFFFF)):
        data = b"BAM\1" + struct.pack("<I", l_text) + b"@HD\n" + struct.pack("<i", n_ref) + struct.pack("<I", 5) + b"chr1\0" + struct.pack("<I", 1000)
        with open(p, "wb") as f:
            f.write(W.bgzf_block(data) + W.BGZF_EOF)
        t0 = time.time()
        with pytest.raises(BscError, match="header"):
            BamReader(p)
        assert time.time() - t0 < 2.0
    sam = str(tmp_path / "p.sam")
    head = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:100000\n"
    for pos, pnext, tlen in (("2147483648", "0", "0"), ("100", "4294967296", "0"), ("-5", "0", "0"), ("100", "0", "-2147483648"), ("99999999999999999999", "0", "0")):
        open(sam, "w").write(head + "r1\t0\tchr1\t%s\t60\t10M\t*\t%s\t%s\tACGTACGTAC\t*\n" % (pos, pnext, tlen))
        with pytest.raises(BscError, match="out of range"):
            c_blocks(sam)
    open(sam, "w").write(head + "r1\t0\tchr1\t2147483647\t60\t10M\t*\t0\t0\tACGTACGTAC\t*\n")  # the largest POS there is: no wrap
    try:
        c_blocks(sam)
    except BscError:
        pass
    W.write_bam(p, REFS, [rec("a", 99, 0x7FFFFFFF, 0x7FFFFFFF, tlen=10)])  # a record at pos 2^31 - 1: read or refused, not undefined
    try:
        c_blocks(p)
    except BscError:
        pass


def test_cigar_that_disagrees_with_the_sequence_length(tmp_path):
    """A CIGAR whose query length is not l_seq: htslib (un-vendored; the reference reads through it) refuses it where it
    parses SAM text (sam_parse1) and hands a binary BAM record over unchecked — the reader errors on the text and, for BAM,
    drops the record before anything walks outside its bases and reads the file on."""
    good = [rec("a", 0, 100, -1), rec("c", 0, 300, -1)]
    bad = rec("b", 0, 200, -1, seq="ACGTACGTAC", cigar=[("M", 14)])
    p = str(tmp_path / "x.bam")
    W.write_bam(p, REFS, [good[0], bad, good[1]])
    blocks, cts, _ = c_blocks(p)
    with BamReader(p, threads=THREADS[0]) as r:  # ... but not unseen: the reader counts what it dropped
        assert r.malformed() == 0
        list(r.blocks())
        assert r.malformed() == 1
    W.write_bam(p, REFS, good)
    assert (blocks, cts) == c_blocks(p)[:2]  # as if the record were not there
    with BamReader(p, threads=THREADS[0]) as r:
        list(r.blocks())
        assert r.malformed() == 0
    sam = str(tmp_path / "x.sam")
    W.write_sam(sam, REFS, [good[0], bad, good[1]])
    with pytest.raises(BscError, match="CIGAR covers 14 query bases"):
        c_blocks(sam)


# ---- random files: C == Python restatement ---------------------------------------------------------------------------------
def _random_records(rng, n):
    recs = []
    for i in range(n):
        tid = int(rng.integers(0, 2))
        pos = int(rng.integers(0, 3000)) if rng.random() < 0.9 else int(rng.integers(10_000, 12_000))
        L = int(rng.integers(20, 60))
        seq = "".join(rng.choice(list("ACGTN"), L, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
        qual = [int(v) for v in rng.integers(2, 60, L)]
        cigar = [("M", L)]
        if rng.random() < 0.3:
            a = int(rng.integers(2, L - 6))
            cigar = [("M", a), (("I", "D")[int(rng.integers(0, 2))], int(rng.integers(1, 4))), ("M", L - a)]
            if cigar[1][0] == "I":
                cigar[2] = ("M", L - a - cigar[1][1])
        if rng.random() < 0.2:
            cigar = [("S", 2)] + cigar
            cigar[1] = ("M", cigar[1][1] - 2)
        tag = [W.aux_char("XB", "C"), W.aux_char("XB", "G"), b"", W.aux_str("XG", "CT")][int(rng.integers(0, 4))]
        mapq = int(rng.choice([0, 19, 20, 30, 60]))
        kind = rng.random()
        name = "r%05d" % (i if rng.random() < 0.97 else max(0, i - 1))
        if kind < 0.55:  # a proper pair
            span = sum(n_ for op, n_ in cigar if op in "MD")
            ins = int(rng.integers(span, 400))
            r1 = bool(rng.integers(0, 2))
            dupf = 1024 if rng.random() < 0.03 else 0
            recs.append(rec(name, 1 | 2 | 32 | (64 if r1 else 128) | dupf, pos, pos + ins - 30, seq, cigar, mapq, tid, tlen=ins, qual=qual, aux=tag))
            seq2 = "".join(rng.choice(list("ACGT"), 30))
            recs.append(rec(name, 1 | 2 | 16 | (128 if r1 else 64) | dupf, pos + ins - 30, pos, seq2, None, int(rng.choice([19, 30, 60])), tid, tlen=-ins,
                            qual=[int(v) for v in rng.integers(2, 60, 30)], aux=tag))
        elif kind < 0.85:
            recs.append(rec(name, int(rng.choice([0, 16, 0, 16, 4, 256, 512, 1024])), pos, -1, seq, cigar, mapq, tid, qual=qual, aux=tag))
        else:  # odd pairs: improper, mate elsewhere, wrong order
            flag = int(rng.choice([1 | 32 | 64, 1 | 2 | 32 | 64, 1 | 2 | 16 | 128, 1 | 8 | 64]))
            recs.append(rec(name, flag, pos, pos + int(rng.integers(-200, 2500)), seq, cigar, mapq, tid, mtid=int(rng.integers(0, 2)),
                            tlen=int(rng.integers(-1500, 1500)), qual=qual, aux=tag))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    return recs


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, {"keep_unmatched": True}), (3, {"keep_duplicates": True}), (4, {"ignore_duplicates": True, "mapq_thresh": 0}),
                                     (5, {"max_template_len": 200})])
def test_c_equals_python_on_random_bams(tmp_path, seed, kw):
    rng = np.random.default_rng(seed)
    n_blocks = n_tpl = 0
    for trial in range(6):
        recs = _random_records(rng, 400)
        p = str(tmp_path / "r.bam")
        W.write_bam(p, REFS, recs, block=int(rng.choice([0xFF00, 777, 4096])))
        try:
            py = py_blocks(p, **kw)
        except AssertionError:
            # the reference asserts (mates that disagree, a mate opening a block, a repeated name): the C reader reports it
            with pytest.raises(BscError):
                c_blocks(p, **kw)
            continue
        c = c_blocks(p, **kw)
        assert c == py, (seed, trial)
        n_blocks += len(c[0])
        n_tpl += sum(len(ts) for _, _, ts in c[0])
    assert n_blocks > 10 and n_tpl > 500


def test_fasta_contig_and_block_reference(tmp_path):
    """load_sequence's alphabet (ACGT either case -> 1..4, anything else printable -> 0, white space skipped), a record found by
    its name up to the first blank, gzip input; get_sequence_string's bounds: the contig's last position reads as N."""
    import gzip

    from bs_call_amd.bam import block_reference, fasta_contig

    fa = tmp_path / "ref.fa"
    fa.write_text(">chr1 first\nACGTNacgtn\nRYKM-ACGT\n>chr10\nTTTT\n>chr2\tsecond\nGGGCCC\n")
    assert fasta_contig(fa, "chr1").tolist() == [1, 2, 3, 4, 0, 1, 2, 3, 4, 0, 0, 0, 0, 0, 0, 1, 2, 3, 4]
    assert fasta_contig(fa, "chr2").tolist() == [3, 3, 3, 2, 2, 2] and fasta_contig(fa, "chr10").tolist() == [4, 4, 4, 4]
    gz = tmp_path / "ref.fa.gz"
    with gzip.open(gz, "wb") as f:
        f.write(fa.read_bytes())
    assert fasta_contig(gz, "chr2", length_hint=2).tolist() == [3, 3, 3, 2, 2, 2]  # a buffer too small is retried with the length
    with pytest.raises(BscError, match="no sequence"):
        fasta_contig(fa, "chr3")
    codes = np.array([3, 3, 3, 2, 2, 2], dtype=np.uint8)
    assert block_reference(codes, 1, 4).tolist() == [3, 3, 3, 2, 2, 0]  # positions 1 .. 6: the 6th (= end_pos) reads 0
    assert block_reference(codes, 4, 8).tolist() == [2, 2, 0, 0, 0, 0, 0]


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_threaded_reader(tmp_path, threads):
    """Helper threads that inflate the BGZF blocks ahead of the parser: the same blocks, counts and errors as without, on
    files of many small BGZF blocks, of one block, truncated files, and readers closed half way."""
    rng = np.random.default_rng(40 + threads)
    THREADS[0] = threads
    try:
        for trial in range(4):
            recs = _random_records(rng, 500)
            p = str(tmp_path / "t.bam")
            W.write_bam(p, REFS, recs, block=int(rng.choice([300, 5000, 0xFF00])))
            try:
                py = py_blocks(p)
            except AssertionError:
                with pytest.raises(BscError):
                    c_blocks(p)
                continue
            assert c_blocks(p) == py
            # closed half way: no hang, no leak of helper threads
            r = BamReader(p, threads=threads)
            it = r.blocks()
            next(it)
            r.close()
            # truncated at a random place: an error (or, cut exactly between blocks, a shorter file), never a hang
            raw = open(p, "rb").read()
            cut = str(tmp_path / "cut.bam")
            open(cut, "wb").write(raw[: int(rng.integers(100, len(raw) - 30))])
            try:
                c_blocks(cut)
            except BscError:
                pass
    finally:
        THREADS[0] = 0


def test_sam_text_input(tmp_path):
    """The same alignments as SAM text — plain and in BGZF blocks — give the blocks, counts and header of the BAM file
    (the line parser re-encodes every alignment as a BAM record in front of the same reader), and the independent
    Python parser of SAM text agrees."""
    rng = np.random.default_rng(2024)
    for trial in range(3):
        recs = _random_records(rng, 400)
        for r in recs:  # names that are unique per template keep the reference's asserts out of this test
            pass
        bam, sam, samz = str(tmp_path / "a.bam"), str(tmp_path / "a.sam"), str(tmp_path / "a.sam.gz")
        W.write_bam(bam, REFS, recs)
        W.write_sam(sam, REFS, recs)
        W.write_sam(samz, REFS, recs, bgzf=True, block=3000)
        try:
            want = c_blocks(bam)
        except BscError:
            for p in (sam, samz):
                with pytest.raises(BscError):
                    c_blocks(p)
            continue
        assert c_blocks(sam) == want and c_blocks(samz) == want
        with BamReader(sam) as r1, BamReader(bam) as r2:
            assert r1.refs == r2.refs == REFS and r1.header_text == r2.header_text
        text, refs, precs = py_bam.parse_sam(sam)
        st = {"cts": [0] * 15, "bases": [0] * 15}
        pyb = [(tid, y, len(als)) for tid, y, als in py_bam.read_input(precs, stats=st)]
        assert pyb == [(tid, y, len(ts)) for tid, y, ts in want[0]] and st["cts"] == want[1]
    # odd but legal text: lower-case bases, '*' qualities, a tag of every kind, no trailing newline, CR LF
    p = str(tmp_path / "odd.sam")
    open(p, "w").write("@HD\tVN:1.6\r\n@SQ\tSN:chr1\tLN:100000\r\n@CO\tfree text\r\n"
                       "r1\t0\tchr1\t101\t60\t2S8M\t*\t0\t0\tacgtnACGTN\t*\tNM:i:3\tXB:A:G\tXX:f:1.5\tYY:Z:hello\tZZ:B:i,1,2,3\tHH:H:1AE3")
    blocks, cts, _ = c_blocks(p)
    t = blocks[0][2][0]
    assert t["bs_strand"] == 2 and t["pos"] == [101, 0] and t["misms"][0] == [[3, 0, 2]]
    assert t["reads"][0] == [b("A", 43), b("C", 43), b("G", 43), b("T", 43), 0, b("A", 43), b("C", 43), b("G", 43), b("T", 43), 0]
    with BamReader(p) as r:
        assert r.refs == [("chr1", 100000)] and r.header_text.count("\n") == 3
    open(p, "w").write("not a sam file at all")
    with pytest.raises(BscError):
        BamReader(p)


def test_region(tmp_path):
    """One region (-r contig:start-stop): the reader sees exactly the alignments an index query would hand it — those of the
    contig that overlap the interval, a read ending AT start - 1 or starting AT stop + 1 excluded, deletions counted in the
    extent — and nothing else is counted in the filter statistics."""
    recs = [rec("a", 0, 89, -1), rec("b", 0, 90, -1), rec("c", 0, 95, -1, cigar=[("M", 3), ("D", 20), ("M", 7)]), rec("d", 0, 200, -1),
            rec("e", 0, 201, -1), rec("f", 512, 150, -1), rec("g", 0, 150, -1, tid=1)]
    p = str(tmp_path / "r.bam")
    W.write_bam(p, REFS, sorted(recs, key=lambda r: (r["tid"], r["pos"])))
    # region chr1:101-201: a covers 90..99 (out), b covers 91..100 (out: ends at start - 1), c covers 96..125 (in), d 201..210 (in),
    # e 202.. (out), f is inside and filtered (QC flag): counted
    blocks, cts, _ = c_blocks(p, region=(0, 101, 201))
    assert [[t["pos"][0] for t in ts] for _, _, ts in blocks] == [[96], [201]] and cts[2] == 1 and sum(cts) == 1
    assert (blocks, cts) == py_blocks(p, region=(0, 101, 201))[:2]
    rng = np.random.default_rng(77)
    n_cmp = 0
    for trial in range(6):
        rr = _random_records(rng, 500)
        W.write_bam(p, REFS, rr)
        reg = (int(rng.integers(0, 2)), int(rng.integers(1, 2500)), 0)
        reg = (reg[0], reg[1], reg[1] + int(rng.integers(1, 1500)))
        try:
            want = py_blocks(p, region=reg)
        except AssertionError:
            with pytest.raises(BscError):
                c_blocks(p, region=reg)
            continue
        assert c_blocks(p, region=reg) == want
        n_cmp += 1
    assert n_cmp >= 3
