"""The CPU oracle chain over one block of reads, for the GPU tests that otherwise compare the HIP path with the product's own host C:
oracle/py_prep.py (pre-processing) -> oracle/bscall_oracle.c accumulate -> call (libm flavour where the host's libm is glibc's FMA
variant, the bsmath.h twin otherwise) -> the print_vcf restatement -> oracle/py_bcf.py (encoder).  No product code in it."""
import numpy as np

import bs_call_amd as B
from oracle import loader as O
from oracle import py_bcf, py_prep


def prepared_arrays(prepared):
    """py_prep.prepare's templates as the arrays the oracle's accumulate takes"""
    tpl = np.zeros(len(prepared), dtype=B.TEMPLATE)
    seq = []
    for i, t in enumerate(prepared):
        tpl["pos"][i] = t["pos"]
        tpl["mapq"][i] = t["mapq"]
        tpl["orientation"][i] = t["orientation"]
        tpl["bs_strand"][i] = t["bs_strand"]
        for k in range(2):
            tpl["off"][i, k] = len(seq)
            tpl["len"][i, k] = len(t["reads"][k])
            seq += t["reads"][k]
    return tpl, np.array(seq, dtype=np.uint8)


def prepare(templates, **kw):
    """raw templates (tests/test_prep.py's dicts) -> (TEMPLATE[], read bytes, the pre-processing's statistics) by py_prep alone"""
    prepared, st = py_prep.prepare(templates, **kw)
    tpl, seq = prepared_arrays(prepared)
    return tpl, seq, st


def records(tpl, seq, x, y, ref, min_qual=20, all_positions=False, dbsnp=None, tables=None):
    """prepared reads -> (VCF_CORE[] of the written records, GT_METH[] of the same positions)"""
    O.build()
    rc, pile = O.accumulate(tpl, seq, x, y, min_qual)
    assert rc == 0
    gtm, skip = O.call_sites(pile, ref[: y - x + 1], tables or O.Tables(), O.LIBM if O.libm_exact() else O.BSM, 1)
    core = O.vcf_block(gtm, skip, ref, x, all_positions=all_positions, dbsnp=dbsnp)
    sel = core["emit"] == 1
    return core[sel], gtm[sel]


def record_dict(c, g):
    return dict(pos=int(c["pos"]), gt=int(c["gt"]), flt=int(c["flt"]), phred=int(c["phred"]), alt=bytes(c["alt"]).rstrip(b"\0"), ref=bytes(c["cx_ref"])[2:3],
                cx_ref=bytes(c["cx_ref"]), cx_gt=bytes(c["cx_gt"]), cg=bytes(c["cg"]), gt_enc=int(c["gt_enc"]), dp=int(c["dp"]), mq=int(g["mq"]), qd=int(c["qd"]),
                fs=int(c["fs"]), gl=[float(v) for v in c["gl"][: int(c["n_gl"])]], counts=[int(v) for v in g["counts"]], qual=[int(v) for v in g["qual"]])


def bcf_stream(core, gtm, rid, name_of=None):
    """the records' BCF2 bytes by the Python encoder; name_of: position -> the dbSNP name (bytes) or None"""
    out = []
    for c, g in zip(core, gtm):
        rs = (name_of(int(c["pos"])) or b"") if name_of else b""
        out.append(py_bcf.encode_record(record_dict(c, g), rid, rs))
    return b"".join(out)


def same_records(recs, core, gtm):
    """packed records of the product (VCF_REC[]) against the chain's"""
    assert len(recs) == len(core)
    assert recs["core"].tobytes() == core.tobytes()
    assert (recs["counts"] == gtm["counts"]).all() and (recs["mq"] == gtm["mq"]).all() and (recs["qual"] == gtm["qual"]).all()
