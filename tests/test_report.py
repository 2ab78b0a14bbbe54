"""csrc/report.c (bsc_report_json) against the independent restatement of output_stats() in oracle/py_report.py, byte for
byte, on populated, sparse and empty statistics; the populated report parses as JSON and carries the numbers."""
import json

import numpy as np
import pytest

from bs_call_amd import report
from bs_call_amd.abi import SITE_STATS
from oracle import py_report

FIELDS = ("snps", "indels", "multi", "dbSNP_sites", "dbSNP_var", "CpG_ref", "CpG_nonref")


def _random_stats(rng, dense):
    t = np.zeros(1, dtype=SITE_STATS)[0]
    for f in FIELDS:
        t[f] = rng.integers(0, 10**12, 2)
    t["mut_counts"] = rng.integers(0, 10**9, (12, 2))
    t["dbSNP_mut_counts"] = rng.integers(0, 10**6, (12, 2))
    t["qual"] = rng.integers(0, 10**10, (4, 256)) * (rng.random((4, 256)) < dense)
    t["filter_counts"] = rng.integers(0, 10**8, (2, 32))
    for f in ("qd_stats", "fs_stats", "mq_stats"):
        t[f] = rng.integers(0, 10**7, (256, 2)) * (rng.random((256, 2)) < dense)
    cov = rng.integers(0, 10**9, (4096, 6)) * (rng.random((4096, 6)) < dense * 0.05)
    t["cov"] = cov
    t["CpG_ref_meth"] = rng.random((2, 101)) * 10 ** rng.integers(-12, 9, (2, 101)).astype(float)
    t["CpG_nonref_meth"] = rng.random((2, 101)) * 1e5
    t["CpG_nonref_meth"][0, :5] = [0.0, 1.0, 123456789.0, 1e-5, 0.1]
    return t


def _as_dict(t, gc, read_profile, contigs, filter_cts, filter_bases, base_filter):
    cov = {}
    for c in range(4096):
        row = [int(v) for v in t["cov"][c]]
        if any(row):
            cov[c] = {"all": row[0], "var": row[1], "CpG": [row[2], row[3]], "CpG_inf": [row[4], row[5]],
                      "gc": [int(v) for v in gc[c]] if gc is not None else [0] * 101}
    d = {f: [int(v) for v in t[f]] for f in FIELDS}
    d.update(
        fs=t["fs_stats"].tolist(), qd=t["qd_stats"].tolist(), mq=t["mq_stats"].tolist(), filter_counts=t["filter_counts"].tolist(),
        qual=t["qual"].tolist(), mut=t["mut_counts"].tolist(), dbsnp_mut=t["dbSNP_mut_counts"].tolist(), cov=cov,
        meth={"ref": t["CpG_ref_meth"].tolist(), "nonref": t["CpG_nonref_meth"].tolist()},
        read_profile=[] if read_profile is None else read_profile.tolist(),
        filter_cts=list(filter_cts) + [0] * (15 - len(filter_cts)), filter_bases=list(filter_bases) + [0] * (15 - len(filter_bases)),
        base_filter=list(base_filter) + [0] * (5 - len(base_filter)),
        contigs=[(n, {f: [int(v) for v in tt[k]] for k, f in enumerate(FIELDS)}) for n, tt in contigs],
    )
    return d


@pytest.mark.parametrize("seed,dense,dbsnp", [(1, 0.9, True), (2, 0.1, False), (3, 0.0, True)])
def test_report_equals_restatement(seed, dense, dbsnp):
    rng = np.random.default_rng(seed)
    t = _random_stats(rng, dense)
    gc = rng.integers(0, 10**6, (4096, 101)).astype(np.uint64) if seed != 2 else None
    prof = rng.integers(0, 10**9, (152, 4)).astype(np.uint64) if seed == 1 else (np.zeros((1, 4), np.uint64) if seed == 3 else None)
    contigs = [("chr%d" % i, rng.integers(0, 10**9, (7, 2)).astype(np.uint64) * (i != 2)) for i in range(1, 5)] if seed != 3 else []
    fcts = [10**9, 0, 5, 7, 0, 11, 0, 0, 3, 0, 0, 0, 13, 0, 2] if seed == 1 else [42]
    fbases = [10**11, 0, 500, 700, 0, 1100, 0, 0, 300, 0, 0, 0, 1300, 0, 200] if seed == 1 else [4200]
    bflt = [10**11, 5, 0, 7, 9] if seed == 1 else [99]
    got = report.render_json(t, 0.01, 0.05, 20, 20, (3, 10, 2026), dbsnp, fcts, fbases, bflt, gc, prof, contigs)
    exp = py_report.output_stats(_as_dict(t, gc, prof, contigs, fcts, fbases, bflt), 0.01, 0.05, 20, 20, (3, 10, 2026), dbsnp)
    assert got == exp
    if seed == 1:
        d = json.loads(got)  # the reference's text is valid JSON whenever no object is left without entries
        assert d["totalStats"]["SNPS"]["All"] == int(t["snps"][0])
        assert d["filterStats"]["ReadLevel"]["LowMAPQ"] == {"Reads": 13, "Bases": 1300}
        assert len(d["totalStats"]["methylation"]["NonCpGreadProfile"]) == 151
        assert list(d["contigStats"]) == ["chr1", "chr3", "chr4"]  # chr2 has no written record: not listed
        assert d["totalStats"]["VCFFilterStats"]["q20,fs60"] == {"NonVariant": int(t["filter_counts"][0][5]), "Variant": int(t["filter_counts"][1][5])}
        cov0 = next(c for c in range(4096) if t["cov"][c][0])
        assert d["totalStats"]["coverage"]["GC"][str(cov0)] == [int(v) for v in gc[cov0]]


def test_report_text_of_a_tiny_run():
    """A few lines typed by hand from the reference's format strings (src/stats.c:29-57, :91-92, :232, :242-245)."""
    t = np.zeros(1, dtype=SITE_STATS)[0]
    t["snps"] = [7, 5]
    t["filter_counts"][0][0], t["filter_counts"][1][0], t["filter_counts"][1][3] = 4, 2, 1
    t["mut_counts"][5] = [3, 2]
    t["CpG_ref_meth"][0][:3] = [0.5, 0.25, 1234567.891]
    t["cov"][30] = [6, 7, 0, 0, 0, 0]
    s = report.render_json(t, 0.01, 0.05, 20, 13, (1, 2, 2025), False, [3], [300], [290, 10])
    assert s.startswith('{\n\t"source": "bs_call_v2.1, under_conversion=0.01, over_conversion=0.05, mapq_thresh=20, bq_thread=13",\n'
                        '\t"date": "01/02/2025",\n\t"filterStats": {\n\t\t"ReadLevel": {\n\t\t\t"Passed": {\n\t\t\t\t"Reads": 3,\n'
                        '\t\t\t\t"Bases": 300\n\t\t\t}\n\t\t},\n\t\t"BaseLevel": {\n\t\t\t"Passed": 290,\n\t\t\t"Trimmed": 10\n\t\t}\n\t},\n'
                        '\t"totalStats": {\n\t\t"SNPS": {\n\t\t\t"All": 7,\n\t\t\t"Passed": 5\n\t\t},\n')
    assert '\n\t\t\t}\n\t\t},\t\t"VCFFilterStats": {\n\t\t\t"PASS": {"NonVariant": 4, "Variant": 2},\n\t\t\t"q20": {"NonVariant": 0, "Variant": 0},\n' in s
    assert '\t\t\t"q20,qd2": {"NonVariant": 0, "Variant": 1},\n' in s
    assert '\t\t\t"C>T": { "All": 3, "Passed": 2, "dbSNPAll": 0, "dbSNPPassed": 0 },\n' in s
    assert '\t\t\t"AllRefCpg": [\n\t\t\t\t0.5, 0.25, 1234567.9, 0, ' in s
    assert '\t\t\t"All": {\n\t\t\t\t"30": 6\n\t\t\t},\n\t\t\t"Variant": {\n\t\t\t\t"30": 7\n\t\t\t},\n\t\t\t"RefCpG": \n\t\t\t},\n' in s
    assert s.endswith('\n\t\t}\n\t},\n\t"contigStats": \n\t}\n}\n')


def test_report_sizing_and_arguments():
    import ctypes as C

    from bs_call_amd import _lib

    L = _lib.load()
    assert L.bsc_report_json(None, None, 0) == -1
    t = np.zeros(1, dtype=SITE_STATS)
    r = _lib.Report()
    r.total = t.ctypes.data
    r.year, r.month, r.day = 2026, 10, 3
    need = L.bsc_report_json(C.byref(r), None, 0)
    small = C.create_string_buffer(100)
    assert L.bsc_report_json(C.byref(r), small, 100) == need and small.raw[99:100] == b"\0" and small.raw[:2] == b"{\n"
