"""Hand-worked records of the reference's VCF record formation (tests/golden/vcf_kav.json, written by
tools/make_vcf_kav.py: every expected value a literal derived by hand from the cited line of src/print_vcf.c) through
  * oracle/py_vcf.py  — the independent pure-Python restatement (the reference's literal tables, its window state machine),
  * oracle/orc_vcf.c  — the C restatement the GPU record kernels are checked against,
  * the GPU record kernel (bsc_vcf_records; -m gpu),
and the two restatements against each other on random blocks (every field of every record)."""
import json
import os

import numpy as np
import pytest

from bs_call_amd.abi import GT_METH, VCF_CORE
from oracle import py_vcf

HERE = os.path.dirname(os.path.abspath(__file__))
KAV = json.load(open(os.path.join(HERE, "golden", "vcf_kav.json")))


def _block(case):
    n = len(case["sites"])
    gtm = np.zeros(n, dtype=GT_METH)
    skip = np.zeros(n, dtype=np.uint8)
    for i, s in enumerate(case["sites"]):
        skip[i] = s["skip"]
        if not s["skip"]:
            gtm["counts"][i] = s["counts"]
            gtm["qual"][i] = s["qual"]
            gtm["gt_prob"][i] = s["gt_prob"]
            gtm["fisher_strand"][i] = s["fisher_strand"]
            gtm["mq"][i], gtm["aq"][i], gtm["max_gt"][i] = s["mq"], s["aq"], s["max_gt"]
    ref = np.array(case["ref"], dtype=np.uint8)
    db = None if case["dbsnp"] is None else np.array(case["dbsnp"], dtype=np.uint8)
    return gtm, skip, ref, db


def _check_record(case, pos, want, got):
    """got: a record dict (py_vcf) or None; want: the hand-worked fields."""
    name = "%s @%s (%s)" % (case["name"], pos, case["decided_by"])
    if want is None:
        assert got is None, name + ": a record was formed"
        return
    assert got is not None, name + ": no record"
    for f, v in want.items():
        if f == "all_positions":
            continue
        if f == "gl":
            assert got["gl"] == [py_vcf._f32(x) for x in v], (name, f, got["gl"])
        else:
            assert got[f] == v, (name, f, got[f], v)


def _core_to_dict(c):
    if int(c["pos"]) == 0:
        return None
    return {"pos": int(c["pos"]), "emit": int(c["emit"]), "gt": int(c["gt"]), "ref_code": int(c["ref_code"]), "gt_enc": int(c["gt_enc"]),
            "flt": int(c["flt"]), "phred": int(c["phred"]), "n_gl": int(c["n_gl"]), "cg": c["cg"].decode(), "alt": c["alt"].decode(),
            "cx_ref": c["cx_ref"].decode(), "cx_gt": c["cx_gt"].decode(), "fs": int(c["fs"]), "qd": int(c["qd"]), "dp": int(c["dp"]),
            "gl": [float(v) for v in c["gl"][: int(c["n_gl"])]]}


def test_fixture_is_what_the_generator_writes():
    assert len(KAV["cases"]) >= 24 and sum(len(c["expect"]) for c in KAV["cases"]) >= 40
    # every FILTER bit, mac1 for each heterozygous genotype, every reference base for GL, every CG state, both flushed positions
    seen_flt = {v.get("flt") for c in KAV["cases"] for v in c["expect"].values() if v}
    assert {0, 1, 2, 4, 8, 11, 128} <= seen_flt
    seen_cg = {v.get("cg") for c in KAV["cases"] for v in c["expect"].values() if v}
    assert {"C", "H", "N", "?", "."} <= seen_cg


@pytest.mark.parametrize("case", KAV["cases"], ids=[c["name"][:40] for c in KAV["cases"]])
def test_python_restatement_on_hand_worked_records(case):
    gtm, skip, ref, db = _block(case)
    recs = py_vcf.vcf_block(gtm, skip, ref, case["x"], case["all_positions"], case["reg_start"], case["reg_stop"], db)
    for pos, want in case["expect"].items():
        _check_record(case, pos, want, recs.get(int(pos)))


@pytest.mark.parametrize("case", KAV["cases"], ids=[c["name"][:40] for c in KAV["cases"]])
def test_c_restatement_on_hand_worked_records(oracle, case):
    gtm, skip, ref, db = _block(case)
    out = oracle.vcf_block(gtm, skip, ref, case["x"], case["all_positions"], case["reg_start"], case["reg_stop"], db)
    for pos, want in case["expect"].items():
        _check_record(case, pos, want, _core_to_dict(out[int(pos) - case["x"]]))


def _random_block(rng, n):
    gtm = np.zeros(n, dtype=GT_METH)
    skip = (rng.random(n) < 0.08).astype(np.uint8)
    for i in range(n):
        if skip[i]:
            continue
        depth = int(rng.choice([1, 2, 4, 30, 300]))
        c = rng.integers(0, depth + 1, 8) * (rng.random(8) < 0.4)
        if not c.any():
            c[rng.integers(0, 8)] = 1
        gtm["counts"][i] = c
        gtm["qual"][i] = np.where(c > 0, rng.integers(20, 44, 8), 0)
        gp = -rng.random(10) * rng.choice([0.01, 1.0, 50.0, 400.0])
        gp[rng.integers(0, 10)] = rng.choice([0.0, -1e-9, -0.001, -0.02, -0.3])
        if rng.random() < 0.1:
            gp[rng.integers(0, 10)] = gp.max()  # a tie: the first maximum is the call
        gtm["gt_prob"][i] = gp
        gtm["fisher_strand"][i] = rng.choice([0.0, 0.0, -0.3, -5.95, -6.05, -20.0])
        gtm["mq"][i] = rng.choice([60, 60, 40, 39, 0])
        gtm["max_gt"][i] = int(np.argmax(gp))
    ref = rng.choice([0, 1, 2, 3, 4], n + 2, p=[0.04, 0.24, 0.24, 0.24, 0.24]).astype(np.uint8)
    db = rng.choice([0, 1, 3], n, p=[0.9, 0.05, 0.05]).astype(np.uint8)
    return gtm, skip, ref, db


def test_two_restatements_agree_on_random_blocks(oracle):
    rng = np.random.default_rng(77)
    for trial in range(60):
        n = int(rng.choice([1, 2, 3, 4, 5, 6, 9, 40, 300]))
        gtm, skip, ref, db = _random_block(rng, n)
        allp = bool(rng.random() < 0.3)
        x = int(rng.choice([1, 2, 3, 5, 1000]))
        reg = (1, 0xFFFFFFFF) if rng.random() < 0.7 else (x + n // 3, x + (2 * n) // 3)
        use_db = db if rng.random() < 0.5 else None
        c_out = oracle.vcf_block(gtm, skip, ref, x, allp, reg[0], reg[1], use_db)
        py = py_vcf.vcf_block(gtm, skip, ref, x, allp, reg[0], reg[1], use_db)
        for i in range(n):
            core = c_out[i]
            rec = py.get(x + i)
            if rec is None:
                assert int(core["pos"]) == 0 and not core.tobytes().strip(b"\0"), (trial, i)
            else:
                assert py_vcf.same_as_core(rec, core) == [], (trial, i, py_vcf.same_as_core(rec, core), rec)


@pytest.mark.gpu
def test_gpu_record_kernel_on_hand_worked_records():
    import bs_call_amd as B

    with B.SiteCaller() as c:
        for case in KAV["cases"]:
            gtm, skip, ref, db = _block(case)
            out = c.vcf_records(gtm, skip, ref, case["x"], case["all_positions"], case["reg_start"], case["reg_stop"], db)
            for pos, want in case["expect"].items():
                _check_record(case, pos, want, _core_to_dict(out[int(pos) - case["x"]]))
