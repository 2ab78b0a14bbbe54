"""Host-side text rendering of bsc_vcf_core records (+ their gt_meth) as VCF data lines — through the library's ONE
formatter, bsc_vcf_format / bsc_vcf_format_rec (host C, csrc/vcf_format.c).

The reference hands each record to htslib (bcf_write, src/print_vcf.c:160-380); htslib is not part of this
repository, so the lines are the layout htslib's VCF text writer gives those fields.  Integer fields are exact by
construction; the float text format of GL (htslib prints floats with %g-style 6 significant digits) is NOT pinned
against htslib here.  No computation happens in this module: every number comes from the device records."""
from .abi import GENOTYPES

FLT_NAMES = ("q20", "qd2", "fs60", "mq40")  # src/init_param.c:15
CS_STR = tuple(("+" if "C" in g else "") + ("-" if "G" in g else "") or "NA" for g in GENOTYPES)  # src/print_vcf.c:61-62
HEADER = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s"


def format_block(cores, gtms, contig):
    """VCF data lines of one block, in position order (records with emit == 0 produce nothing): bsc_vcf_format."""
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


format_block_c = format_block  # round-1 name
