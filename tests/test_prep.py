"""Read pre-processing (csrc/prep.c, host C) against hand-worked cases — one per branch of
trim_read / trim_soft_clips / handle_overlap / the indel normalisation, each derived by hand from the cited lines of the
reference — and against the pure-Python restatement (oracle/py_prep.py) on random valid alignments.
Pinning caveat (same as the rest of the oracle): the reference's own code for this stage cannot be built in this image;
the hand-worked cases are the anchor.  CPU only: this stage is host code."""
import numpy as np
import pytest

import bs_call_amd as B
from bs_call_amd.abi import MISMS, RAW_TEMPLATE
from bs_call_amd.caller import prepare_templates
from oracle import py_prep

INS, DEL, SOFT = 1, 2, 3


def b(base, q=30):
    return base | (q << 2)


def read(n, start=0, q=30):
    """n bytes whose bases cycle A C G T from `start` (so that every position is recognisable)."""
    return [b((start + i) & 3, q) for i in range(n)]


def tpl(pos, span, reads, misms=((), ()), mapq=(60, 60), orientation=0, bs_strand=1):
    return {"pos": list(pos), "span": list(span), "reads": [None if r is None else list(r) for r in reads],
            "misms": [[list(m) for m in misms[0]], [list(m) for m in misms[1]]], "mapq": list(mapq), "orientation": orientation,
            "bs_strand": bs_strand}


def to_arrays(templates):
    raw = np.zeros(len(templates), dtype=RAW_TEMPLATE)
    seq, ms = [], []
    for i, t in enumerate(templates):
        raw["pos"][i] = t["pos"]
        raw["reference_span"][i] = t["span"]
        raw["mapq"][i] = t["mapq"]
        raw["orientation"][i] = t["orientation"]
        raw["bs_strand"][i] = t["bs_strand"]
        for k in range(2):
            r = t["reads"][k]
            raw["off"][i, k] = len(seq)
            raw["len"][i, k] = 0 if r is None else len(r)
            seq += [] if r is None else r
            raw["misms_off"][i, k] = len(ms)
            raw["n_misms"][i, k] = len(t["misms"][k])
            ms += [tuple(m) for m in t["misms"][k]]
    return raw, np.array(seq, dtype=np.uint8), np.array(ms, dtype=MISMS) if ms else np.zeros(0, dtype=MISMS)


def run_c(templates, **kw):
    raw, seq, ms = to_arrays(templates)
    out, oseq, st = prepare_templates(raw, seq, ms, **kw)
    res = []
    for o in out:
        res.append({"pos": [int(v) for v in o["pos"]], "reads": [oseq[int(o["off"][k]) : int(o["off"][k]) + int(o["len"][k])].tolist() for k in range(2)],
                    "mapq": [int(v) for v in o["mapq"]], "orientation": int(o["orientation"]), "bs_strand": int(o["bs_strand"])})
    return res, {f: int(st[f]) for f in st.dtype.names}


def both(templates, **kw):
    try:
        c, cst = run_c(templates, **kw)
    except B.BscError as e:
        # the one input class where the reference itself is undefined (a memmove with a negative count): both report it
        assert "indel beyond the read" in str(e)
        with pytest.raises(py_prep.PrepError):
            py_prep.prepare(templates, **kw)
        return None, None
    p, pst = py_prep.prepare(templates, **kw)
    assert c == p and cst == pst
    return c, cst


# ---- hand-worked cases -------------------------------------------------------------------------------------------------
def test_fixed_trims_and_the_right_trim_quirk():
    """src/read_utils.c:13-26: left trim marks the first bases q = 63 keeping their base; the right trim writes the LAST
    bases but takes the base from sp[k1], i.e. from the LEFT end.  -L/-R index read 1 / read 2: on a REVERSE template
    read[0] is read 2 (src/process_template.c:37-41)."""
    r = read(10)  # bases 0 1 2 3 0 1 2 3 0 1
    out, st = both([tpl((100, 0), (10, 0), (r, None))], left_trim=(2, 0), right_trim=(3, 0))
    F = 63 << 2
    assert out[0]["reads"][0] == [0 | F, 1 | F] + r[2:7] + [2 | F, 1 | F, 0 | F]  # sp[9] <- base of sp[0], sp[8] <- sp[1], sp[7] <- sp[2]
    assert st["base_trim"] == 5 and st["base_none"] == 5
    # REVERSE template: the read-1 trims apply to read[1]
    out, _ = both([tpl((100, 300), (10, 10), (read(10), read(10, 1)), orientation=1)], left_trim=(1, 0), right_trim=(0, 0))
    assert out[0]["reads"][0] == read(10) and out[0]["reads"][1][0] == (1 | F) and out[0]["reads"][1][1:] == read(10, 1)[1:]


def test_soft_clips():
    """src/al_utils.c:122-162: a leading clip is cut from the left and every later mismatch position shifts by it; a
    trailing clip is cut from the right; the clip entries leave the list."""
    r = read(20)
    # 3S 10M 2D(reference's INS at read offset 13) 4M 3S : misms SOFT@0/3, INS@13/2, SOFT@17/3
    t = tpl((100, 0), (16, 0), (r, None), (((SOFT, 0, 3), (INS, 13, 2), (SOFT, 17, 3)), ()))
    out, st = both([t])
    # after clips: r[3:17] (14 bytes), INS now at offset 10 -> two 0 bytes inserted there: 16 bytes = the reference span
    assert out[0]["reads"][0] == r[3:13] + [0, 0] + r[13:17] and st["base_clip"] == 6
    # a clip in the middle of the list, or one that swallows the read: the reference aborts; here an error naming the template
    for bad in ((((INS, 3, 1), (SOFT, 5, 2), (INS, 9, 1)), ()), (((SOFT, 0, 20),), ()), (((SOFT, 5, 10),), ())):
        raw, seq, ms = to_arrays([tpl((100, 0), (20, 0), (r, None), bad)])
        with pytest.raises(B.BscError) as e:
            prepare_templates(raw, seq, ms)
        assert "template 0" in str(e.value)
        with pytest.raises(py_prep.PrepError):
            py_prep.prepare([tpl((100, 0), (20, 0), (r, None), bad)])


def test_overlap_equal_spans_quality_decides_right_trim_of_read0():
    """src/al_utils.c:170-232: fwd at 100 span 20, rev at 110 span 20: overlap = 20 - 110 + 100 = 10.  Equal spans: mean
    quality; read 0 has the lower (or equal) mean -> trim_read = 0; !rev && !trim_read -> RIGHT trim of read 0 by 10."""
    r0, r1 = read(20, 0, 25), read(20, 2, 35)
    out, st = both([tpl((100, 110), (20, 20), (r0, r1))])
    assert out[0]["reads"] == [r0[:10], r1] and out[0]["pos"] == [100, 110] and st["base_overlap"] == 10
    # equal means: tot[0] <= tot[1] still trims read 0
    out, _ = both([tpl((100, 110), (20, 20), (read(20, 0, 30), read(20, 2, 30)))])
    assert [len(x) for x in out[0]["reads"]] == [10, 20]


def test_overlap_longer_span_wins_left_trim_moves_the_start():
    """rspan[0] > rspan[1] -> trim_read = 1; not rev and trim_read -> LEFT trim of read 1 and reverse_position += overlap
    (:204-208): fwd 100 span 30, rev 120 span 20: overlap = 30 - 120 + 100 = 10 -> read 1 loses its first 10, starts at 130."""
    r0, r1 = read(30), read(20, 1)
    out, st = both([tpl((100, 120), (30, 20), (r0, r1))])
    assert out[0]["reads"] == [r0, r1[10:]] and out[0]["pos"] == [100, 130] and st["base_overlap"] == 10


def test_overlap_reverse_read_first():
    """forward_position > reverse_position: rev = true, overlap = span[1] + rev - fwd (:174-177).  rev 100 span 25,
    fwd 115 span 20: overlap = 25 + 100 - 115 = 10.  rspan[0] (20) < rspan[1] (25) -> trim_read = 0; rev && !trim_read is
    neither right-trim case -> LEFT trim of read 0, forward_position += 10."""
    r0, r1 = read(20), read(25, 3)
    out, _ = both([tpl((115, 100), (20, 25), (r0, r1))])
    assert out[0]["reads"] == [r0[10:], r1] and out[0]["pos"] == [125, 100]
    # rev && trim_read: spans 30 / 20 -> trim_read = 1: RIGHT trim of read 1, no position change
    out, _ = both([tpl((115, 100), (30, 20), (read(30), read(20, 3)))])  # overlap = 20 + 100 - 115 = 5
    assert out[0]["reads"] == [read(30), read(20, 3)[:15]] and out[0]["pos"] == [115, 100]


def test_no_overlap_and_single_reads_are_left_alone():
    out, st = both([tpl((100, 200), (20, 20), (read(20), read(20, 1))), tpl((100, 0), (20, 0), (read(20), None)),
                    tpl((0, 150), (0, 20), (None, read(20, 2)))])
    assert [x["reads"] for x in out] == [[read(20), read(20, 1)], [read(20), []], [[], read(20, 2)]] and st["base_overlap"] == 0
    assert st["reads"] == 4 and st["read_bases"] == 80
    # abutting mates (fwd + span == rev) count as overlapping by 0: nothing is cut (:181, overlap = 0)
    out, _ = both([tpl((100, 120), (20, 20), (read(20), read(20, 1)))])
    assert out[0]["reads"] == [read(20), read(20, 1)]


def test_overlap_right_trim_walks_the_indels():
    """Right trim with a mismatch list (:217-241): read 0 = 10M 2D 10M (read length 20, span 22, INS@10/2), rev at 112:
    overlap = 22 - 112 + 100 = 10, xx = span - overlap = 12.  z = 0: position 10 + adj 0 < 12; INS: 10 + 0 + 2 >= 12 ->
    trim = rdl - position = 10, size = xx - (position + adj) = 2, list keeps this entry.  Read 0 = first 10 bases, then the
    normalisation pads the 2 deleted positions: 12 bytes covering 100..111."""
    r0, r1 = read(20, 0, 20), read(24, 1, 40)
    out, _ = both([tpl((100, 112), (22, 24), (r0, r1), (((INS, 10, 2),), ()))])
    assert out[0]["reads"][0] == r0[:10] + [0, 0] and out[0]["reads"][1] == r1
    # the cut falls before the indel: entry position >= xx -> trim = rdl - xx + adj, the list is cut at that entry
    out, _ = both([tpl((100, 106), (22, 24), (r0, r1), (((INS, 10, 2),), ()))])  # overlap 16, xx = 6
    assert out[0]["reads"][0] == r0[:6]
    # an insertion (reference's DEL) before the cut shifts read offsets: 5M 3I 12M, read 20, span 17; rev at 110 ->
    # overlap 7, xx = 10; DEL@5/3: adj = -3; no entry reaches xx -> plain right trim by the overlap (7): 13 bytes left, the
    # normalisation removes the 3 inserted bases: 10 bytes covering 100..109
    r2 = read(20, 2, 20)
    out, _ = both([tpl((100, 110), (17, 24), (r2, r1), (((DEL, 5, 3),), ()))])
    assert out[0]["reads"][0] == r2[:5] + r2[8:13]


def test_overlap_left_trim_walks_the_indels():
    """Left trim with a mismatch list (:242-301): read 1 (trimmed: shorter span) = 6M 2D 12M (length 18, span 20, INS@6/2)
    at 120; fwd at 100 span 30: overlap = 10, xx = 10.  z = 0: position 6 < 10; INS: 6 + 0 + 2 = 8 < 10 -> adj = 2.  No
    entry reaches xx -> left trim by overlap - adj = 8, list emptied, reverse_position = 130."""
    r0, r1 = read(30), read(18, 1)
    out, _ = both([tpl((100, 120), (30, 20), (r0, r1), ((), ((INS, 6, 2),)))])
    assert out[0]["reads"] == [r0, r1[8:]] and out[0]["pos"] == [100, 130]
    # the cut falls before the indel: entry position >= xx -> trim = overlap - adj, later entries shift by it
    r1 = read(28, 1)  # 15M 2D 13M: span 30 ... make read 0 longer
    out, _ = both([tpl((100, 130), (40, 30), (read(40), r1), ((), ((INS, 15, 2),)))])  # overlap = 10: cut 10, INS now at 5
    assert out[0]["reads"][1] == r1[10:15] + [0, 0] + r1[15:] and out[0]["pos"] == [100, 140]
    # the cut falls INSIDE a deletion: 4M 10D 10M (length 14, span 24, INS@4/10) at 120, overlap 8: 4 + 0 + 10 >= 8 ->
    # size = 4 + 10 + 0 - 8 = 6, trim = position = 4: the read keeps its last 10 bases behind 6 padded positions
    r1 = read(14, 2)
    out, _ = both([tpl((100, 120), (28, 24), (read(28), r1), ((), ((INS, 4, 10),)))])
    assert out[0]["reads"][1] == [0] * 6 + r1[4:] and out[0]["pos"] == [100, 128]


def test_indel_normalisation_alone():
    """src/process_template.c:86-105: INS (deletion from the reference) inserts `size` bytes 0 at position + adj; DEL
    (insertion) removes `size` bytes there."""
    r = read(12)
    out, _ = both([tpl((100, 0), (13, 0), (r, None), (((INS, 3, 2), (DEL, 7, 1)), ()))])
    assert out[0]["reads"][0] == r[:3] + [0, 0] + r[3:7] + r[8:]


def test_get_al_qual_uses_sq_k():
    """src/al_utils.c:19-35: q = GET_QUAL(sq[k]) with k the READ index: read 0 is scored by its first byte, read 1 by its
    second — rl times each."""
    t = tpl((100, 200), (4, 3), ([b(0, 10), b(1, 40), b(2, 40), b(3, 40)], [b(0, 40), b(1, 20), b(2, 40)]))
    raw, seq, _ = to_arrays([t])
    L = __import__("bs_call_amd._lib", fromlist=["load"]).load()
    got = L.bsc_template_qual(raw.ctypes.data, seq.ctypes.data)
    assert got == py_prep.get_al_qual(t) == (4 * 10 + 3 * 20) // 7


# ---- random valid alignments: C == Python restatement -----------------------------------------------------------------------
def _random_read(rng, qlo=5):
    """A CIGAR [S] M (D|I M)* [S] -> (read bytes, mismatch list, reference span)."""
    ops = []
    if rng.random() < 0.3:
        ops.append(("S", int(rng.integers(1, 6))))
    ops.append(("M", int(rng.integers(8, 40))))
    for _ in range(int(rng.integers(0, 4))):
        ops.append((("D", "I")[int(rng.integers(0, 2))], int(rng.integers(1, 5))))
        ops.append(("M", int(rng.integers(3, 30))))
    if rng.random() < 0.3:
        ops.append(("S", int(rng.integers(1, 6))))
    pos = span = 0
    ms = []
    for op, n in ops:
        if op == "M":
            pos += n
            span += n
        elif op == "S":
            ms.append([SOFT, pos, n])
            pos += n
        elif op == "I":
            ms.append([DEL, pos, n])
            pos += n
        else:
            ms.append([INS, pos, n])
            span += n
    rd = [int(rng.integers(0, 4)) | (int(rng.integers(qlo, 44)) << 2) for _ in range(pos)]
    return rd, ms, span


def test_c_equals_python_on_random_alignments():
    rng = np.random.default_rng(424242)
    for trial in range(300):
        ts = []
        for _ in range(int(rng.integers(1, 12))):
            r0, m0, s0 = _random_read(rng)
            r1, m1, s1 = _random_read(rng)
            kind = rng.random()
            p0 = int(rng.integers(50, 5000))
            if kind < 0.15:
                t = tpl((p0, 0), (s0, 0), (r0, None), (m0, ()))
            elif kind < 0.3:
                t = tpl((0, p0), (0, s1), (None, r1), ((), m1))
            else:
                p1 = p0 + int(rng.integers(-s1 - 5, s0 + 30))
                t = tpl((p0, max(1, p1)), (s0, s1), (r0, r1), (m0, m1))
            t["orientation"] = int(rng.integers(0, 2))
            t["bs_strand"] = int(rng.integers(0, 3))
            ts.append(t)
        lt = tuple(int(v) for v in rng.integers(0, 4, 2)) if rng.random() < 0.3 else (0, 0)
        rt = tuple(int(v) for v in rng.integers(0, 4, 2)) if rng.random() < 0.3 else (0, 0)
        both(ts, left_trim=lt, right_trim=rt, min_qual=int(rng.integers(1, 44)))


def test_prepared_templates_feed_the_accumulate_stage(oracle):
    """What comes out is what a-3 consumes: the oracle's accumulate over the prepared templates (CPU; the GPU twin of
    this check is tests/test_gpu_accumulate.py::test_prepared_templates_on_the_device)."""
    rng = np.random.default_rng(9)
    ts = []
    for i in range(200):
        r0, m0, s0 = _random_read(rng, 20)
        r1, m1, s1 = _random_read(rng, 20)
        p0 = 1000 + 7 * i
        ts.append(tpl((p0, p0 + int(rng.integers(0, s0 + 20))), (s0, s1), (r0, r1), (m0, m1), orientation=int(rng.integers(0, 2))))
    raw, seq, ms = to_arrays(ts)
    out, oseq, _ = prepare_templates(raw, seq, ms)
    x = 998
    y = int(max((o["pos"][k] + o["len"][k]) for o in out for k in range(2) if o["len"][k])) + 1
    rc, pile = oracle.accumulate(out, oseq, x, y, 20)
    assert rc == 0 and int(pile["n"].sum()) > 0
    # padded deletions (byte 0: q = 0) never count; every other base with q >= 20 counts once
    want = sum(1 for o in out for k in range(2) for c in oseq[int(o["off"][k]) : int(o["off"][k]) + int(o["len"][k])] if (c >> 2) >= 20 and (c >> 2) != 63)
    assert int(pile["n"].sum()) == want


def test_prepared_templates_say_whether_read0_was_walked():
    """bsc_template.flags (round 5): bsc_prepare_templates and the L-reads generator hand over whether read 0 holds a base whose
    quality is neither 0 nor 63 — what the scan of src/call_genotypes.c:198-211 decides for the orientation flip of :224 — so
    that the device stage need not fetch it.  Against a plain numpy scan, on inputs that have fully trimmed read 0s."""
    from bs_call_amd.reads import TPL_WALK_KNOWN, TPL_WALKED0, walk_flags

    rng = np.random.default_rng(77)
    ts = []
    for i in range(400):
        r0, m0, s0 = _random_read(rng, 30)
        r1, m1, s1 = _random_read(rng, 30)
        p0 = 2000 + 5 * i
        kind = i % 4
        if kind == 0:
            r0 = [b(c & 3, 63 if j % 2 else 0) for j, c in enumerate(r0)]  # nothing countable in read 0
        ts.append(tpl((p0, p0 + int(rng.integers(0, s0 + 40))) if kind != 1 else (0, p0), (s0, s1) if kind != 1 else (0, s1),
                      (r0, r1) if kind != 1 else (None, r1), (m0, m1) if kind != 1 else ((), m1)))
    raw, seq, ms = to_arrays(ts)
    out, oseq, _ = prepare_templates(raw, seq, ms, left_trim=(3, 0), right_trim=(0, 2))
    want = walk_flags(out, oseq)
    assert (out["flags"] == want).all()
    assert (out["flags"] == TPL_WALK_KNOWN).sum() >= 100 and (out["flags"] == (TPL_WALK_KNOWN | TPL_WALKED0)).sum() >= 100
    gt, gs = B.synth_reads_host(88172645463325252, 1000, 200_000, 30)
    assert (gt["flags"] == walk_flags(gt, gs)).all() and (gt["flags"] == TPL_WALK_KNOWN).any()


# ---- the non-CpG read profile (meth_profile, src/meth_profile.c) -----------------------------------------------------------
def _profile_both(templates, x, ref, **kw):
    from bs_call_amd.caller import ReadProfile

    raw, seq, ms = to_arrays(templates)
    pf = ReadProfile(cap=4096)
    prepare_templates(raw, seq, ms, profile=pf, x=x, ref=np.array(ref, dtype=np.uint8), **kw)
    pp = py_prep.Profile(4096)
    py_prep.prepare(templates, profile=pp, x=x, ref=list(ref), **kw)
    assert pf.used == pp.used and pf.counts.tolist() == pp.mem
    return pf


def test_read_profile_hand_worked():
    """Reference (codes 1..4 = ACGT) from x = 10:  A C A C G T G G C T  (positions 10 .. 19) + 2 more.
    Non-CpG cytosines: 11 (C then A), 18 (C then T); 13 is CpG.  Guanines not after a C: 16 (T G), 17 (G G); 14 is CpG.
    A C2T forward read at 11 showing  T A C G T G A C T  (quality 30):
      pos 11 ref C, read T -> flt_tab[C2T][T] = 5: count 1 (b... index 1), a C / T observation at a non-CpG C: counted
      pos 13 ref C (CpG), read C = 4: index 0, but the reference context is CpG: not counted
      pos 16 ref G, read G = 10 = 8 | 2: index 2, a G / A observation at a G not preceded by C: counted
      pos 17 ref G, read A = 11 = 8 | 3: index 3: counted
      pos 18 ref C, read C = 4: index 0: counted
    Read position i is element i + 1."""
    ref = [1, 2, 1, 2, 3, 4, 3, 3, 2, 4, 1, 1]
    T_, A_, C_, G_ = 3, 0, 1, 2
    rd = [b(v) for v in (T_, A_, C_, G_, T_, G_, A_, C_, T_)]
    t = tpl((11, 0), (9, 0), (rd, None), bs_strand=1)
    pf = _profile_both([t], 10, ref)
    assert pf.used == 10  # max_pos = trim_left + rl = 9
    exp = np.zeros((4096, 4), dtype=np.uint64)
    exp[1 + 0][1] = 1  # pos 11
    exp[1 + 5][2] = 1  # pos 16
    exp[1 + 6][3] = 1  # pos 17
    exp[1 + 7][0] = 1  # pos 18
    assert (pf.counts == exp).all()
    # the same bases as the reverse read of a template: positions count from the far end (posx - k1), and a low-quality base
    # (q < 20) has no table entry
    rd2 = list(rd)
    rd2[0] = b(T_, 10)
    t2 = tpl((0, 11), (0, 9), (None, rd2), bs_strand=1)
    pf2 = _profile_both([t2], 10, ref)
    assert pf2.used == 9  # max_pos = rl + trim_right - 1 = 8: one less than for a forward read
    exp2 = np.zeros((4096, 4), dtype=np.uint64)
    exp2[1 + 8 - 5][2] = 1
    exp2[1 + 8 - 6][3] = 1
    exp2[1 + 8 - 7][0] = 1
    assert (pf2.counts == exp2).all()


def test_read_profile_soft_clip_deletion_and_the_block_start_quirk():
    """A left soft clip shifts the positions by its length; a deletion's padding lands in element 0; a read starting AT the
    block start is walked one reference base late (state starts at 0 and the pointer is not advanced, :66-67)."""
    ref = [2, 1, 2, 4, 3, 1, 2, 1, 4, 4, 1, 1, 1, 1]  # x = 1:  C A C T G A C A T T ...
    C_, T_ = 1, 3
    # two clipped bases, then C T | one reference base deleted | T C, aligned at position 3
    rd = [b(0), b(0), b(C_), b(T_), b(T_), b(C_)]
    t = tpl((3, 0), (5, 0), (rd, None), (([SOFT, 0, 2], [INS, 4, 1]), ()), bs_strand=1)
    pf = _profile_both([t], 1, ref)
    # after the clip: read C T T C at orig 2..5, deletion padded at index 2 -> C T 0 T C over positions 3 4 5 6 7 (ref C T G A C)
    # pos 3 ref C followed by T: read C (4, index 0) counted at orig 2; pos 7 ref C followed by A: read C counted at orig 5
    assert pf.used == 2 + 4 + 1 and pf.counts[1 + 2][0] == 1 and pf.counts[1 + 5][0] == 1 and int(pf.counts.sum()) == 2
    # block start: the read begins at x itself
    t3 = tpl((1, 0), (4, 0), ([b(C_), b(0), b(C_), b(T_)], None), bs_strand=1)
    pf3 = _profile_both([t3], 1, ref)
    # walked one base late: at base j the state is (ref[j-1], ref[j]) instead of (ref[j], ref[j+1]), so the non-CpG C at
    # position 1 (C then A) is credited to base 1 (an A: a G / A observation, no match) and the C at 3 (C then T) to base 3,
    # the T (5 = C / T observation, index 1): one count, at read position 3, where a correct walk would give C->C at 0 and 2
    exp3 = np.zeros((4096, 4), dtype=np.uint64)
    exp3[1 + 3][1] = 1
    assert (pf3.counts == exp3).all()


def test_read_profile_c_equals_python_on_random_alignments():
    from bs_call_amd.caller import ReadProfile

    rng = np.random.default_rng(777)
    ref = [int(v) for v in rng.integers(0, 5, 6000)]
    pf = ReadProfile(cap=1024)
    pp = py_prep.Profile(1024)
    n_done = 0
    for trial in range(200):
        ts = []
        for _ in range(int(rng.integers(1, 8))):
            r0, m0, s0 = _random_read(rng, qlo=10)
            r1, m1, s1 = _random_read(rng, qlo=10)
            p0 = int(rng.integers(5, 5000))
            kind = rng.random()
            if kind < 0.2:
                t = tpl((p0, 0), (s0, 0), (r0, None), (m0, ()))
            elif kind < 0.4:
                t = tpl((0, p0), (0, s1), (None, r1), ((), m1))
            else:
                t = tpl((p0, max(5, p0 + int(rng.integers(-s1 - 5, s0 + 30)))), (s0, s1), (r0, r1), (m0, m1))
            t["orientation"] = int(rng.integers(0, 2))
            t["bs_strand"] = int(rng.integers(0, 3))
            ts.append(t)
        x = max(1, min(min(p for p in t["pos"] if p) for t in ts) - 2)
        raw, seq, ms = to_arrays(ts)
        import copy

        trial_pp = copy.deepcopy(pp)
        try:
            py_prep.prepare(ts, profile=trial_pp, x=x, ref=ref[x - 1 :])
        except py_prep.PrepError:
            continue  # the reference-undefined class: the batch is skipped on both sides
        pp = trial_pp
        prepare_templates(raw, seq, ms, profile=pf, x=x, ref=np.array(ref[x - 1 :], dtype=np.uint8))
        assert pf.used == pp.used and pf.counts.tolist() == pp.mem, trial
        n_done += 1
    assert n_done > 150 and int(pf.counts.sum()) > 1000 and int(pf.counts[0].sum()) == 0  # padding never counts (byte 0 has no entry)
