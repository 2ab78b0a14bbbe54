"""The C oracle (statement-list restatement) against oracle/py_model.py (matrix-form restatement, pure Python):
bit-for-bit agreement on random sites, plus the SURVEY 8c vectors through the Python model."""
import json
import os
import random

import numpy as np

from oracle import py_model

HERE = os.path.dirname(os.path.abspath(__file__))


def test_python_model_reproduces_reference_vectors():
    kav = json.load(open(os.path.join(HERE, "golden", "kav_survey8c.json")))
    for c in kav["calc_gt_prob"]:
        mx, gp = py_model.calc_gt_prob(c["counts"], c["qual"], c["rf"])
        assert mx == c["max_gt"] and gp[mx] == float.fromhex(c["gt_prob_max_hex"]), c["name"]
        for name, val in c["gt_prob"].items():
            assert gp[py_model.GENOTYPES.index(name)] == val
    import math

    store = [0.0, 0.0]
    l = 0.0
    for i in range(2, 256):
        l += math.log(float(i))
        store.append(l)
    for f in kav["fisher"]:
        assert py_model.fisher(f["c"], store) == float.fromhex(f["p_hex"])


def test_two_restatements_agree(oracle, tables):
    rng = random.Random(20240)
    store = list(tables.lfact_store)
    for trial in range(4000):
        depth = rng.choice([1, 2, 5, 30, 200])
        counts = [rng.randrange(0, depth + 1) if rng.random() < 0.45 else 0 for _ in range(8)]
        if not any(counts):
            counts[rng.randrange(8)] = 1
        quals = [rng.randrange(20, 44) if c else 0 for c in counts]
        rf = rng.randrange(0, 5)
        params = rng.choice([(0.01, 0.05, 2.0), (0.0, 0.0, 1.0), (0.2, 0.1, 5.0)])
        tb = tables if params == (0.01, 0.05, 2.0) else oracle.Tables(*params)
        g = oracle.calc_gt_prob(counts, quals, rf, tb, oracle.LIBM)
        mx, gp = py_model.calc_gt_prob(counts, quals, rf, *params)
        assert mx == int(g["max_gt"]), (counts, quals, rf, params)
        assert gp == [float(v) for v in g["gt_prob"]], (counts, quals, rf, params)
    for trial in range(3000):
        hi = rng.choice([3, 20, 120, 600])
        c = [rng.randrange(0, hi) for _ in range(4)]
        assert py_model.fisher(c, store) == oracle.fisher(c, tables, oracle.LIBM), c
