"""The LOGIC of the device BAM reader (csrc/bamdev_core.h: record -> descriptor, block segmentation, pair table, duplicate resolution,
template assembly) against csrc/bamio.c and oracle/py_bam.py, on the CPU: tests/emul/bamdev_emul.cpp compiles the header's statements
with g++ and runs them by loops where csrc/bamdev.hip runs them by lanes (tests/test_gpu_bamdev.py checks the kernels themselves).
Every scenario of tests/test_bam.py and its random files go through (a) the sequential replay alone, (b) the parallel path with the
replay for irregular input, each in one pass and in passes of a few records (blocks carried over a pass's end)."""
import ctypes
import gzip
import importlib.util
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("test_bam_mod", os.path.join(ROOT, "tests", "test_bam.py"))
TB = importlib.util.module_from_spec(spec)
spec.loader.exec_module(TB)
W = TB.W

RAW_T = np.dtype([("pos", "<u4", 2), ("reference_span", "<u4", 2), ("len", "<u4", 2), ("n_misms", "<u4", 2), ("off", "<u8", 2), ("misms_off", "<u8", 2),
                  ("mapq", "u1", 2), ("orientation", "u1"), ("bs_strand", "u1"), ("_pad", "<u4")])
MISMS_T = np.dtype([("type", "<u4"), ("position", "<u4"), ("size", "<u4")])


class Params(ctypes.Structure):
    _fields_ = [("mapq_thresh", ctypes.c_uint32), ("keep_unmatched", ctypes.c_uint32), ("ignore_duplicates", ctypes.c_uint32), ("keep_duplicates", ctypes.c_uint32),
                ("max_template_len", ctypes.c_uint64), ("region_tid", ctypes.c_int32), ("region_start", ctypes.c_uint32), ("region_stop", ctypes.c_uint32),
                ("n_ref", ctypes.c_int32), ("keep_unplaced", ctypes.c_uint32), ("tid_keep", ctypes.c_void_p)]


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("emul") / "libbamdev_emul.so")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-Wall", "-o", out, os.path.join(ROOT, "tests", "emul", "bamdev_emul.cpp")])
    lib = ctypes.CDLL(out)
    lib.bd_emul_error.restype = ctypes.c_char_p
    lib.bd_emul_n_blocks.restype = ctypes.c_uint64
    return lib


def stream_of(path):
    """the inflated file, the offsets of its records, n_ref"""
    raw = gzip.open(path, "rb").read()
    assert raw[:4] == b"BAM\1"
    o = 8 + struct.unpack_from("<I", raw, 4)[0]
    n_ref = struct.unpack_from("<i", raw, o)[0]
    o += 4
    for _ in range(n_ref):
        o += 8 + struct.unpack_from("<I", raw, o)[0]
    offs = []
    while o < len(raw):
        offs.append(o)
        o += 4 + struct.unpack_from("<I", raw, o)[0]
    return raw, np.array(offs, dtype=np.uint64), n_ref


class EmulError(Exception):
    pass


def emul_blocks(lib, path, mode=0, pass_recs=0, region=None, mapq_thresh=20, max_template_len=1000, keep_unmatched=False, ignore_duplicates=False,
                keep_duplicates=False, stream=None, contigs=None):
    """stream: (inflated bytes, record offsets, n_ref) instead of the whole file's (a contig selection's stretches); contigs: the record parser's
    contig filter (tids, -1 = the unplaced reads)"""
    raw, offs, n_ref = stream if stream is not None else stream_of(path)
    par = Params(mapq_thresh, int(keep_unmatched), int(ignore_duplicates), int(keep_duplicates), max_template_len, 0, 0, 0, n_ref, 0, None)
    keep = None
    if contigs is not None:
        keep = np.zeros(n_ref + 1, dtype=np.uint8)
        for t in contigs:
            if t >= 0:
                keep[t] = 1
        par.keep_unplaced = int(any(t < 0 for t in contigs))
        par.tid_keep = keep.ctypes.data
    if region:
        par.region_tid, par.region_start, par.region_stop = region
    buf = (ctypes.c_uint8 * len(raw)).from_buffer_copy(raw)
    rc = lib.bd_emul_run(buf, ctypes.c_uint64(len(raw)), offs.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(offs)), ctypes.byref(par),
                         ctypes.c_uint32(pass_recs), ctypes.c_int(mode))
    if rc:
        raise EmulError(lib.bd_emul_error().decode())
    out = []
    for i in range(lib.bd_emul_n_blocks()):
        tid, y, nr = ctypes.c_int32(), ctypes.c_uint32(), ctypes.c_uint32()
        sb, nm = ctypes.c_uint64(), ctypes.c_uint64()
        lib.bd_emul_block(ctypes.c_uint64(i), ctypes.byref(tid), ctypes.byref(y), ctypes.byref(nr), ctypes.byref(sb), ctypes.byref(nm))
        tpl = np.zeros(nr.value, dtype=RAW_T)
        seq = np.zeros(max(1, sb.value), dtype=np.uint8)
        ms = np.zeros(max(1, nm.value), dtype=MISMS_T)
        lib.bd_emul_block_data(ctypes.c_uint64(i), tpl.ctypes.data_as(ctypes.c_void_p), seq.ctypes.data_as(ctypes.c_void_p), ms.ctypes.data_as(ctypes.c_void_p))
        out.append((tid.value, y.value, templates_as_dicts(tpl, seq, ms)))
    cts = (ctypes.c_uint64 * 15)()
    bases = (ctypes.c_uint64 * 15)()
    mal, ur, uf = ctypes.c_uint64(), ctypes.c_int(), ctypes.c_int()
    lib.bd_emul_counts(cts, bases, ctypes.byref(mal), ctypes.byref(ur), ctypes.byref(uf))
    return (out, list(cts), list(bases)), {"malformed": mal.value, "replay": ur.value, "fast": uf.value}


def templates_as_dicts(tpl, seq, ms):
    """the shape tests/test_bam.py compares (c_blocks)"""
    ts = []
    for t in tpl:
        reads, misms = [], []
        for k in range(2):
            ln = int(t["len"][k])
            reads.append(seq[int(t["off"][k]): int(t["off"][k]) + ln].tolist() if ln else None)
            o, n = int(t["misms_off"][k]), int(t["n_misms"][k])
            misms.append([[int(m["type"]), int(m["position"]), int(m["size"])] for m in ms[o: o + n]])
        ts.append({"pos": [int(v) for v in t["pos"]], "span": [int(t["reference_span"][k]) if t["len"][k] else 0 for k in range(2)],
                   "reads": reads, "misms": misms, "mapq": [int(v) for v in t["mapq"]], "orientation": int(t["orientation"]),
                   "bs_strand": int(t["bs_strand"])})
    return ts


VARIANTS = [(1, 0), (1, 7), (0, 0), (0, 5)]  # (mode, records per pass)


def check_file(lib, path, **kw):
    """bamio.c's answer (or its refusal) from every variant of the emulated device reader; returns (blocks, which paths decided)"""
    try:
        want = TB.c_blocks(path, **kw)
    except TB.BscError:
        want = None
    info_all = []
    for mode, pr in VARIANTS:
        if want is None:
            with pytest.raises(EmulError):
                emul_blocks(lib, path, mode=mode, pass_recs=pr, **kw)
            continue
        got, info = emul_blocks(lib, path, mode=mode, pass_recs=pr, **kw)
        assert got == want, (mode, pr)
        info_all.append(info)
    return want, info_all


def test_hand_worked_scenarios_of_test_bam(emul, tmp_path):
    """every scenario of tests/test_bam.py: written by its own code, read by bamio.c and by the emulated device reader"""
    r = TB.rec
    files = {
        "pair": [r("p", 99, 1000, 1200, "ACGTACGTAC", tlen=210, aux=W.aux_char("XB", "C"), mapq=50), r("p", 147, 1200, 1000, "TTTTTGGGGG", tlen=-210, aux=W.aux_char("XB", "C"), mapq=40)],
        "tags": [r("a", 163, 100, 300, "ACNTR", tlen=205, qual=[50, 43, 40, 7, 30], aux=W.aux_str("ZS", "-+")), r("b", 99, 102, 302, "ACGTA", tlen=205, qual="missing", aux=W.aux_str("XG", "GA")),
                 r("c", 0, 104, -1, "ACGTA", aux=W.aux_int("NM", 3) + W.aux_str("YD", "f")), r("d", 16, 106, -1, "ACGTA", aux=W.aux_str("ZB", "CT")),
                 r("a", 83, 300, 100, "GGGGG", tlen=-205, aux=W.aux_str("ZS", "-+")), r("b", 147, 302, 102, "CCCCC", tlen=-205, aux=W.aux_str("XG", "GA"))],
        "cigar": [r("x", 0, 500, -1, "A" * 15, cigar=[("S", 3), ("M", 4), ("I", 2), ("M", 3), ("D", 5), ("M", 2), ("S", 1)])],
        "gaps": [r("a", 0, 100, -1), r("b", 0, 110, -1), r("c", 0, 121, -1), r("d", 0, 133, -1), r("e", 0, 50, -1, tid=1)],
        "dups": [r("p1", 99, 100, 300, tlen=210, mapq=30), r("p2", 99, 100, 300, tlen=210, mapq=50), r("p3", 99, 100, 300, tlen=210, mapq=50, qual=[20] * 10),
                 r("s1", 0, 105, -1, mapq=40), r("s2", 0, 105, -1, mapq=41), r("p1", 147, 300, 100, tlen=-210, mapq=30), r("p2", 147, 300, 100, tlen=-210, mapq=50),
                 r("p3", 147, 300, 100, tlen=-210, mapq=50)],
        "lone_mate": [r("a", 0, 100, -1), r("m", 147, 400, 100, tlen=-310)],
        "bad_cigar": [r("a", 0, 100, -1), r("b", 0, 200, -1, seq="ACGTACGTAC", cigar=[("M", 14)]), r("c", 0, 300, -1)],
        "same_pos_mates": [r("e", 99, 700, 700, tlen=10), r("e", 147, 700, 700, tlen=-10), r("f", 99, 700, 700, tlen=10, mapq=30), r("f", 147, 700, 700, tlen=-10, mapq=30)],
    }
    ok1, ok2 = r("ok", 99, 100, 200, tlen=110), r("ok", 147, 200, 100, tlen=-110)
    bad = [r("sec", 99 | 256, 101, 200), r("unm", 1 | 4, 102, 200), r("mun", 1 | 8, 103, 200), r("qc", 99 | 512, 104, 200), r("dup", 99 | 1024, 105, 200),
           r("npp", 1 | 32 | 64, 106, 200), r("lowq", 99, 107, 200, mapq=19), r("chr", 99, 108, 200, mtid=1), r("long", 99, 109, 2000, tlen=1901),
           r("ori", 99, 210, 110, tlen=-100), r("s_dup", 1024, 111, -1), r("s_qc", 512, 112, -1)]
    files["filters"] = sorted([ok1, ok2] + bad, key=lambda q: q["pos"])
    for name, recs in files.items():
        p = str(tmp_path / (name + ".bam"))
        W.write_bam(p, TB.REFS, recs)
        want, infos = check_file(emul, p)
        if name == "bad_cigar":
            assert all(i["malformed"] == 1 for i in infos)
        if name in ("pair", "gaps", "cigar", "filters", "dups", "same_pos_mates"):
            assert infos[2]["fast"] >= 1 and infos[2]["replay"] == 0, name  # ordinary input: the parallel path decides alone
    kd = [r("p1", 99, 100, 300, tlen=210), r("p2", 99, 100, 300, tlen=210), r("far", 99, 105, 5000, tlen=4905), r("p1", 147, 300, 100, tlen=-210),
          r("p2", 147, 300, 100, tlen=-210)]
    p = str(tmp_path / "kd.bam")
    W.write_bam(p, TB.REFS, kd)
    check_file(emul, p, keep_duplicates=True)
    check_file(emul, p, keep_duplicates=True, keep_unmatched=True)
    reg = [r("a", 0, 89, -1), r("b", 0, 90, -1), r("c", 0, 95, -1, cigar=[("M", 3), ("D", 20), ("M", 7)]), r("d", 0, 200, -1), r("e", 0, 201, -1), r("f", 512, 150, -1),
           r("g", 0, 150, -1, tid=1)]
    W.write_bam(p, TB.REFS, sorted(reg, key=lambda q: (q["tid"], q["pos"])))
    check_file(emul, p, region=(0, 101, 201))


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, {"keep_unmatched": True}), (3, {"keep_duplicates": True}), (4, {"ignore_duplicates": True, "mapq_thresh": 0}),
                                     (5, {"max_template_len": 200}), (6, {"keep_unmatched": True, "keep_duplicates": True})])
def test_random_bams_of_test_bam(emul, tmp_path, seed, kw):
    """tests/test_bam.py's random files — re-used names, odd pairs, negative mate positions among them: every variant gives bamio.c's
    blocks and counts, or refuses where it refuses"""
    rng = np.random.default_rng(seed)
    n_cmp = n_err = 0
    for trial in range(8):
        recs = TB._random_records(rng, 400)
        p = str(tmp_path / "r.bam")
        W.write_bam(p, TB.REFS, recs, block=int(rng.choice([0xFF00, 777, 4096])))
        want, _ = check_file(emul, p, **kw)
        n_cmp += want is not None
        n_err += want is None
    assert n_cmp >= 3


def _sane_records(rng, n, dup_rate=0.1):
    """coordinate-sorted pairs and singles with unique names, duplicates at shared start positions, mates at one position, indels and
    clips: input the parallel path must decide by itself"""
    recs = []
    pos = 0
    for i in range(n):
        if rng.random() > dup_rate:
            pos += int(rng.integers(0, 40)) if rng.random() < 0.97 else int(rng.integers(300, 900))
        tid = 0
        L = int(rng.integers(20, 60))
        seq = "".join(rng.choice(list("ACGTN"), L, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
        qual = [int(v) for v in rng.integers(2, 60, L)]
        cigar = [("M", L)]
        if rng.random() < 0.3:
            a = int(rng.integers(2, L - 6))
            k = int(rng.integers(1, 4))
            cigar = [("M", a), ("I", k), ("M", L - a - k)] if rng.random() < 0.5 else [("M", a), ("D", k), ("M", L - a)]
        if rng.random() < 0.2:
            cigar = [("S", 2)] + cigar
            cigar[1] = ("M", cigar[1][1] - 2)
        tag = [W.aux_char("XB", "C"), W.aux_char("XB", "G"), b"", W.aux_str("XG", "CT")][int(rng.integers(0, 4))]
        mapq = int(rng.choice([19, 20, 30, 60]))
        name = "q%06d" % i
        kind = rng.random()
        if kind < 0.65:
            ins = int(rng.choice([0, 30, 100, 250]))  # 0: both mates at one position
            r1 = bool(rng.integers(0, 2))
            recs.append(TB.rec(name, 1 | 2 | 32 | (64 if r1 else 128), pos, pos + ins, seq, cigar, mapq, tid, tlen=ins + 30, qual=qual, aux=tag))
            seq2 = "".join(rng.choice(list("ACGT"), 30))
            recs.append(TB.rec(name, 1 | 2 | 16 | (128 if r1 else 64), pos + ins, pos, seq2, None, int(rng.choice([19, 30, 60])), tid, tlen=-(ins + 30),
                               qual=[int(v) for v in rng.integers(2, 60, 30)], aux=tag))
        else:
            recs.append(TB.rec(name, int(rng.choice([0, 16])), pos, -1, seq, cigar, mapq, tid, qual=qual, aux=tag))
    # stable: the forward mate of a pair at one position stays in front of its partner
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    return recs


@pytest.mark.parametrize("seed,kw", [(11, {}), (12, {"keep_duplicates": True}), (13, {"keep_unmatched": True}), (14, {"mapq_thresh": 0})])
def test_ordinary_input_never_needs_the_replay(emul, tmp_path, seed, kw):
    rng = np.random.default_rng(seed)
    for trial in range(5):
        recs = _sane_records(rng, 600, dup_rate=float(rng.choice([0.0, 0.1, 0.5])))
        p = str(tmp_path / "s.bam")
        W.write_bam(p, TB.REFS, recs)
        want, infos = check_file(emul, p, **kw)
        assert want is not None and len(want[0]) >= 1
        assert infos[2]["replay"] == 0 and infos[3]["replay"] == 0, (seed, trial)
        py = TB.py_blocks(p, **kw)
        assert want == py
