#!/usr/bin/env python3
"""Writes tests/golden/chain_regression.json: a small block taken through the whole oracle chain (reads -> pile-up ->
gt_meth -> VCF records -> statistics) with its outputs as hex digests plus a few spelled-out records.  NOT a reference
vector — the reference cannot be built here — but an anchor that makes any later change to the oracle (or to the
generators) visible: tests/test_oracle_kav.py recomputes it on the CPU, tests/test_gpu_records.py on the GPU."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bs_call_amd as B
from oracle import loader as O

SEED, X0, N, COV = 20261003, 5000, 3000, 25


def chain():
    tb = O.Tables()
    tpl, seq = B.synth_reads_host(SEED, X0, N, COV)
    x = X0 - 2
    y = int((tpl["pos"] + tpl["len"]).max()) - 1
    ref = B.synth_ref_host(SEED, x, y - x + 3)
    rc, pile = O.accumulate(tpl, seq, x, y, 20)
    gtm, skip = O.call_sites(pile, ref[: y - x + 1], tb, O.BSM, 1)  # bsm flavour: the same bits on every host
    st = np.zeros(1, dtype=B.SITE_STATS)
    core = O.vcf_block_stats(gtm, skip, ref, x, st, np.zeros(2, dtype=np.uint32), tb.lfact_store)
    return tpl, seq, x, y, ref, pile, gtm, skip, core, st[0]


def digest(a):
    """sha256 over the FIELDS of a record array (numpy leaves the padding bytes of a copied record undefined)."""
    a = np.asarray(a)
    h = hashlib.sha256()
    if a.dtype.names:
        for f in a.dtype.names:
            h.update(digest(a[f]).encode())
    else:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def summary():
    tpl, seq, x, y, ref, pile, gtm, skip, core, st = chain()
    ints = {f: st[f].ravel().tolist() for f in ("snps", "multi", "CpG_ref", "CpG_nonref")}
    ints["mut_counts"] = st["mut_counts"].ravel().tolist()
    ints["filter_counts"] = st["filter_counts"].ravel().tolist()
    emit = np.flatnonzero(core["emit"] == 1)
    recs = []
    for i in emit[:: max(1, len(emit) // 12)][:12]:
        c = core[i]
        recs.append({"pos": int(c["pos"]), "gt": int(c["gt"]), "ref_code": int(c["ref_code"]), "flt": int(c["flt"]),
                     "phred": int(c["phred"]), "cg": c["cg"].decode(), "alt": c["alt"].decode(), "fs": int(c["fs"]),
                     "qd": int(c["qd"]), "dp": int(c["dp"]), "counts": gtm["counts"][i].tolist(),
                     "gt_prob_hex": [float(v).hex() for v in gtm["gt_prob"][i]]})
    return {
        "_source": "tools/make_golden_chain.py (oracle, bsm flavour, single thread); regression anchor, not a reference vector",
        "generator": {"seed": SEED, "x0": X0, "n_sites": N, "coverage": COV},
        "block": {"x": int(x), "y": int(y), "templates": int(len(tpl)), "bases": int(len(seq))},
        "sha256": {"templates": digest(tpl), "seq": digest(seq), "ref": digest(ref), "pileup": digest(pile),
                   "gt_meth": digest(gtm), "skip": digest(skip), "vcf_core": digest(core),
                   "site_stats_int": digest(np.frombuffer(st.tobytes()[: B.SITE_STATS_INT_WORDS * 8], dtype=np.uint64))},
        "site_stats": ints,
        "meth_profile_sum": float(st["CpG_ref_meth"][0].sum() + st["CpG_nonref_meth"][0].sum()),
        "records": recs,
    }


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden", "chain_regression.json")
    json.dump(summary(), open(out, "w"), indent=1)
    print("wrote", out)
