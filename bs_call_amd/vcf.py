"""Host-side text rendering of bsc_vcf_core records (+ their gt_meth) as VCF data lines.

The reference hands each record to htslib (bcf_write, src/print_vcf.c:160-380); htslib is not part of this
repository, so this is the layout htslib's VCF text writer gives those fields.  Integer fields are exact by
construction; the float text format of GL (htslib prints floats with %g-style 6 significant digits) is NOT pinned
against htslib here.  No computation happens in this module: every number comes from the device records."""
from .abi import GENOTYPES

FLT_NAMES = ("q20", "qd2", "fs60", "mq40")  # src/init_param.c:15
CS_STR = tuple(("+" if "C" in g else "") + ("-" if "G" in g else "") or "NA" for g in GENOTYPES)  # src/print_vcf.c:61-62
HEADER = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s"


def _gl(v):
    return "%g" % float(v)


def format_record(core, gtm, contig, rs_id="."):
    """One VCF data line for a record with emit == 1 (core: VCF_CORE scalar, gtm: GT_METH scalar)."""
    gt = int(core["gt"])
    het = GENOTYPES[gt][0] != GENOTYPES[gt][1]
    flt = int(core["flt"])
    alt = core["alt"].decode()
    cols = [contig, str(int(core["pos"])), rs_id, core["cx_ref"].decode()[2], ",".join(alt) if alt else ".",
            str(int(core["phred"])), "PASS" if flt == 0 else ("mac1" if flt & 128 else "fail"),
            "CX=" + core["cx_ref"].decode()]
    enc = int(core["gt_enc"])
    a, b = (enc >> 4 >> 1) - 1, ((enc & 15) >> 1) - 1
    ft = ";".join(n for i, n in enumerate(FLT_NAMES) if flt >> i & 1) if flt & 15 else "PASS"
    counts = [int(c) for c in gtm["counts"]]
    amq = [str(int(q)) for c, q in zip(counts, gtm["qual"]) if c > 0]
    keys = ["GT", "FT", "DP", "MQ", "GQ", "QD", "GL", "MC8"]
    vals = ["%d/%d" % (a, b), ft, str(int(core["dp"])), str(int(gtm["mq"])), str(int(core["phred"])), str(int(core["qd"])),
            ",".join(_gl(v) for v in core["gl"][: int(core["n_gl"])]), ",".join(map(str, counts))]
    if amq:
        keys.append("AMQ")
        vals.append(",".join(amq))
    keys += ["CS", "CG", "CX"]
    vals += [CS_STR[gt], core["cg"].decode(), core["cx_gt"].decode()]
    if het:
        keys.append("FS")
        vals.append(str(int(core["fs"])))
    return "\t".join(cols + [":".join(keys), ":".join(vals)])


def format_block(cores, gtms, contig):
    """VCF data lines of one block, in position order (records with emit == 0 produce nothing)."""
    return [format_record(c, g, contig) for c, g in zip(cores, gtms) if c["emit"]]


def format_block_c(cores, gtms, contig):
    """The same lines through the library's C formatter (bsc_vcf_format)."""
    import ctypes as C

    import numpy as np

    from . import _lib
    from .abi import GT_METH, VCF_CORE

    L = _lib.load()
    cores = np.ascontiguousarray(cores, dtype=VCF_CORE)
    gtms = np.ascontiguousarray(gtms, dtype=GT_METH)
    buf = C.create_string_buffer(1024)
    out = []
    cb, gb = cores.ctypes.data, gtms.ctypes.data
    for i in range(len(cores)):
        n = L.bsc_vcf_format(cb + 64 * i, gb + 200 * i, contig.encode(), None, buf, 1024)
        if n < 0:
            raise RuntimeError("bsc_vcf_format: buffer too small")
        if n:
            out.append(buf.raw[:n].decode())
    return out


def format_records_c(recs, contig):
    """VCF lines of packed records (VCF_REC[], what SiteCaller.block_records returns) through bsc_vcf_format_rec."""
    import ctypes as C

    import numpy as np

    from . import _lib
    from .abi import VCF_REC

    L = _lib.load()
    recs = np.ascontiguousarray(recs, dtype=VCF_REC)
    buf = C.create_string_buffer(1024)
    out = []
    rb = recs.ctypes.data
    for i in range(len(recs)):
        n = L.bsc_vcf_format_rec(rb + 128 * i, contig.encode(), None, buf, 1024)
        if n < 0:
            raise RuntimeError("bsc_vcf_format_rec: buffer too small")
        if n:
            out.append(buf.raw[:n].decode())
    return out
