"""The device BAM reader (csrc/bamstream.c + csrc/bamdev.hip, round 6): blocks of raw templates formed in HBM against csrc/bamio.c (host C)
AND oracle/py_bam.py (independent Python restatement) on every scenario and random file of tests/test_bam.py, on ordinary files that the
parallel kernels must decide alone, with the one-lane replay forced, with slabs and passes small enough that records straddle slabs and
blocks are carried over passes; then file to file: pipeline.run(device_reader=True) writes the bytes of the host reader's run."""
import importlib.util
import os

import numpy as np
import pytest

import bs_call_amd as B
from bs_call_amd import pipeline
from bs_call_amd.bamdev import DeviceBamReader
from bs_call_amd.caller import BscError

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


TB = _load("test_bam_mod3", "tests/test_bam.py")
TE = _load("test_bamdev_emul_mod", "tests/test_bamdev_emul.py")
W = TB.W


@pytest.fixture(scope="module")
def caller():
    c = B.SiteCaller()
    yield c
    c.close()


def dev_blocks(caller, path, env=None, **kw):
    """(blocks as tests/test_bam.py compares them, filter counts, filter bases), run statistics"""
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        out = []
        with DeviceBamReader(caller, path, threads=3, **kw) as r:
            for tid, y, tpl, seq, ms in r.blocks():
                out.append((tid, y, TE.templates_as_dicts(tpl, seq, ms)))
            cts, bases = r.filter_counts()
            info = r.run_stats()
            info["malformed"] = r.malformed()
        return (out, cts, bases), info
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


VARIANTS = [({}, "kernels"), ({"BSC_BAMDEV_REPLAY": "1"}, "replay"), ({"BSC_BAMDEV_SLAB_KB": "64", "BSC_BAMDEV_PASS_KB": "1"}, "small passes"),
            ({"BSC_BAMDEV_SLAB_KB": "64", "BSC_BAMDEV_PASS_KB": "1", "BSC_BAMDEV_REPLAY": "1"}, "small passes, replay")]


def check_file(caller, path, variants=VARIANTS, **kw):
    try:
        want = TB.c_blocks(path, **kw)
    except BscError:
        want = None
    infos = []
    for env, what in variants:
        if want is None:
            with pytest.raises(BscError):
                dev_blocks(caller, path, env, **kw)
            continue
        got, info = dev_blocks(caller, path, env, **kw)
        assert got == want, what
        infos.append(info)
    return want, infos


def test_hand_worked_scenarios(caller, tmp_path):
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
        "long_reads": [r("l1", 0, 100, -1, "ACGT" * 75 + "AC", qual=list(range(2, 62)) * 5 + [30, 31]), r("l2", 16, 150, -1, "TTGCA" * 61, qual=[40] * 305)],
    }
    ok1, ok2 = r("ok", 99, 100, 200, tlen=110), r("ok", 147, 200, 100, tlen=-110)
    bad = [r("sec", 99 | 256, 101, 200), r("unm", 1 | 4, 102, 200), r("mun", 1 | 8, 103, 200), r("qc", 99 | 512, 104, 200), r("dup", 99 | 1024, 105, 200),
           r("npp", 1 | 32 | 64, 106, 200), r("lowq", 99, 107, 200, mapq=19), r("chr", 99, 108, 200, mtid=1), r("long", 99, 109, 2000, tlen=1901),
           r("ori", 99, 210, 110, tlen=-100), r("s_dup", 1024, 111, -1), r("s_qc", 512, 112, -1)]
    files["filters"] = sorted([ok1, ok2] + bad, key=lambda q: q["pos"])
    for name, recs in files.items():
        p = str(tmp_path / (name + ".bam"))
        W.write_bam(p, TB.REFS, recs)
        want, infos = check_file(caller, p)
        if name != "bad_cigar":  # (the Python restatement walks that record as the reference would; bamio.c and the device drop and count it)
            assert want == TB.py_blocks(p), name
        if name == "bad_cigar":
            assert all(i["malformed"] == 1 for i in infos)
        if name in ("pair", "gaps", "cigar", "filters", "dups", "same_pos_mates"):
            assert infos[0]["replay_passes"] == 0, name
    kd = [r("p1", 99, 100, 300, tlen=210), r("p2", 99, 100, 300, tlen=210), r("far", 99, 105, 5000, tlen=4905), r("p1", 147, 300, 100, tlen=-210),
          r("p2", 147, 300, 100, tlen=-210)]
    p = str(tmp_path / "kd.bam")
    W.write_bam(p, TB.REFS, kd)
    check_file(caller, p, keep_duplicates=True)
    check_file(caller, p, keep_duplicates=True, keep_unmatched=True)
    reg = [r("a", 0, 89, -1), r("b", 0, 90, -1), r("c", 0, 95, -1, cigar=[("M", 3), ("D", 20), ("M", 7)]), r("d", 0, 200, -1), r("e", 0, 201, -1), r("f", 512, 150, -1),
           r("g", 0, 150, -1, tid=1)]
    W.write_bam(p, TB.REFS, sorted(reg, key=lambda q: (q["tid"], q["pos"])))
    check_file(caller, p, region=(0, 101, 201))
    # an empty file, a header and nothing else
    W.write_bam(p, TB.REFS, [])
    want, _ = check_file(caller, p)
    assert want[0] == []


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, {"keep_unmatched": True}), (3, {"keep_duplicates": True}), (4, {"ignore_duplicates": True, "mapq_thresh": 0}),
                                     (5, {"max_template_len": 200}), (6, {"keep_unmatched": True, "keep_duplicates": True})])
def test_random_bams_of_test_bam(caller, tmp_path, seed, kw):
    """re-used names, odd pairs, negative mate positions: bamio.c's blocks and counts, or its refusal; and py_bam's"""
    rng = np.random.default_rng(seed)
    n_cmp = 0
    for trial in range(5):
        recs = TB._random_records(rng, 400)
        p = str(tmp_path / "r.bam")
        W.write_bam(p, TB.REFS, recs, block=int(rng.choice([0xFF00, 777, 4096])))
        want, _ = check_file(caller, p, variants=VARIANTS[:2] if trial else VARIANTS, **kw)
        if want is not None:
            assert want == TB.py_blocks(p, **kw)
            n_cmp += 1
    assert n_cmp >= 2


@pytest.mark.parametrize("seed,kw", [(11, {}), (12, {"keep_duplicates": True}), (13, {"keep_unmatched": True}), (14, {"mapq_thresh": 0})])
def test_ordinary_input_is_decided_by_the_parallel_kernels(caller, tmp_path, seed, kw):
    """sorted records, names that pair up: no pass may need the replay; 6 000 records in 64-KiB slabs = a dozen passes, blocks carried"""
    rng = np.random.default_rng(seed)
    for trial in range(3):
        recs = TE._sane_records(rng, 6000 if trial == 0 else 800, dup_rate=float(rng.choice([0.0, 0.1, 0.5])))
        p = str(tmp_path / "s.bam")
        W.write_bam(p, TB.REFS, recs)
        want, infos = check_file(caller, p, variants=[VARIANTS[0], VARIANTS[2]], **kw)
        assert want is not None and len(want[0]) >= 1
        assert all(i["replay_passes"] == 0 for i in infos), (seed, trial)
        if trial == 0:
            assert infos[1]["passes"] >= 4
        assert want == TB.py_blocks(p, **kw)


def test_bam_to_bcf_with_the_device_reader(tmp_path):
    """file to file: the device reader in front of the device pre-processing, calling and encoding writes the BCF and report bytes of the
    host reader's run (which tests/test_gpu_pipeline.py checks against the oracle chain)"""
    rng = np.random.default_rng(77)
    reference = {"chrA": rng.integers(1, 5, 30_000).astype(np.uint8), "chrB": rng.integers(1, 5, 12_000).astype(np.uint8)}
    reference["chrA"][5_000:5_400] = 0
    refs = [(k, len(v)) for k, v in reference.items()]
    recs = W.wgbs_records(rng, reference["chrA"], 0, 1500, het_every=500) + W.wgbs_records(rng, reference["chrB"], 1, 500)
    bam = str(tmp_path / "in.bam")
    W.write_bam(bam, refs, recs)
    outs = []
    for tag, kw in (("host", {}), ("dev", {"device_reader": True})):
        bcf, rep = str(tmp_path / (tag + ".bcf")), str(tmp_path / (tag + ".json"))
        res = pipeline.run(bam, reference, bcf, sample="S1", report_path=rep, date=(3, 10, 2026), compressed=False, **kw)
        outs.append((open(bcf, "rb").read(), open(rep).read(), res["blocks"], res["records"]))
    assert outs[0] == outs[1] and outs[0][3] > 5_000
