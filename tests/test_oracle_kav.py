"""Pin the CPU oracle (libm flavour) to the known-answer vectors the reference itself produced
(tests/golden/kav_survey8c.json) and check the bsm flavour against them too."""
import json
import os

import numpy as np
import pytest

from bs_call_amd.abi import GENOTYPES, GT_METH, PILEUP, TEMPLATE

HERE = os.path.dirname(os.path.abspath(__file__))
KAV = json.load(open(os.path.join(HERE, "golden", "kav_survey8c.json")))


def test_struct_sizes(oracle):
    L = oracle.lib()
    assert L.orc_sizeof_pileup() == PILEUP.itemsize == 104
    assert L.orc_sizeof_gt_meth() == GT_METH.itemsize == 200
    assert L.orc_sizeof_template() == TEMPLATE.itemsize == 40


@pytest.mark.parametrize("case", KAV["calc_gt_prob"], ids=lambda c: c["name"])
def test_calc_gt_prob_libm_bit_exact(oracle, tables, case):
    g = oracle.calc_gt_prob(case["counts"], case["qual"], case["rf"], tables, oracle.LIBM)
    assert int(g["max_gt"]) == case["max_gt"]
    assert float(g["gt_prob"][case["max_gt"]]) == float.fromhex(case["gt_prob_max_hex"])
    for name, val in case["gt_prob"].items():
        # SURVEY prints these with 17 significant digits: exact round trip
        assert float(g["gt_prob"][GENOTYPES.index(name)]) == val, name


@pytest.mark.parametrize("case", KAV["calc_gt_prob"], ids=lambda c: c["name"])
def test_calc_gt_prob_bsm(oracle, tables, libm_exact, case):
    """The kernels' arithmetic (bsmath.h replica of glibc log/exp): same bits as the reference's vectors."""
    b = oracle.calc_gt_prob(case["counts"], case["qual"], case["rf"], tables, oracle.BSM)
    assert int(b["max_gt"]) == case["max_gt"]
    assert float(b["gt_prob"][case["max_gt"]]) == float.fromhex(case["gt_prob_max_hex"])
    for name, val in case["gt_prob"].items():
        assert float(b["gt_prob"][GENOTYPES.index(name)]) == val, name


@pytest.mark.parametrize("case", KAV["fisher"], ids=lambda c: "-".join(map(str, c["c"])))
def test_fisher(oracle, tables, case):
    p = oracle.fisher(case["c"], tables, oracle.LIBM)
    assert p == float.fromhex(case["p_hex"])
    if "p" in case:
        assert p == case["p"]
    assert oracle.fisher(case["c"], tables, oracle.BSM) == p


def test_fisher_does_not_leak_mutation(oracle, tables):
    """fisher() mutates its argument in the reference (src/stats_utils.c:50-53); the loader copies."""
    c = [3, 5, 7, 2]
    oracle.fisher(c, tables)
    assert c == [3, 5, 7, 2]


def test_q_prob_table(tables):
    """fill_base_prob_table (src/genotype_model.c:10-21): q=0 -> e clipped to 0.5 -> k = 0.5."""
    qp = tables.q_prob
    assert qp[0, 0] == 0.5 and qp[0, 1] == 0.5
    assert qp[0, 2] == np.log(0.5) and qp[0, 3] == 0.0
    e30 = np.exp(-0.1 * 30 * 2.30258509299404568402)
    assert qp[30, 0] == e30 and qp[30, 1] == e30 / (3.0 - 4.0 * e30)
    assert tables.lfact_store[0] == 0 and tables.lfact_store[1] == 0
    import math

    assert abs(tables.lfact_store[255] - math.lgamma(256.0)) < 1e-9  # ln(255!)


def test_chain_regression_anchor(libm_exact):
    """tests/golden/chain_regression.json (tools/make_golden_chain.py): the whole oracle chain on one small block gives
    the digests recorded when the fixture was written — a change to the oracle or to the generators shows up here."""
    import importlib.util
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_golden_chain", os.path.join(root, "tools", "make_golden_chain.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    gold = json.load(open(os.path.join(root, "tests", "golden", "chain_regression.json")))
    now = m.summary()
    for k in ("templates", "seq", "ref", "pileup", "gt_meth", "skip"):
        assert now["sha256"][k] == gold["sha256"][k], k
    assert now["block"] == gold["block"]
    if libm_exact:  # the printer's phred goes through the host's exp / log
        assert now["sha256"]["vcf_core"] == gold["sha256"]["vcf_core"]
        assert now["sha256"]["site_stats_int"] == gold["sha256"]["site_stats_int"]
        assert now["site_stats"] == gold["site_stats"] and now["records"] == gold["records"]
        assert abs(now["meth_profile_sum"] - gold["meth_profile_sum"]) < 1e-9
