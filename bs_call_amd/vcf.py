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


# ---- BCF output: what the reference's output file holds (src/print_vcf.c:621-731 the header, :160-380 the records) -----------
INFO_FILTER_FORMAT_LINES = (  # src/print_vcf.c:712-731, verbatim texts (the FS line's missing '>' included)
    '##INFO=<ID=CX,Number=1,Type=String,Description="5 base sequence context (from position -2 to +2 on the positive strand) determined from the reference">',
    '##FILTER=<ID=fail,Description="No sample passed filters">',
    '##FILTER=<ID=q20,Description="Genotype Quality below 20">',
    '##FILTER=<ID=qd2,Description="Quality By Depth below 2">',
    '##FILTER=<ID=fs60,Description="Fisher Strand above 60">',
    '##FILTER=<ID=mq40,Description="RMS Mapping Quality below 40">',
    '##FILTER=<ID=mac1,Description="Minor allele count <= 1">',
    '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
    '##FORMAT=<ID=FT,Number=1,Type=String,Description="Sample Genotype Filter">',
    '##FORMAT=<ID=GL,Number=G,Type=Float,Description="Genotype Likelihood">',
    '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Phred scaled conditional genotype quality">',
    '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Read Depth (non converted reads only)">',
    '##FORMAT=<ID=MQ,Number=1,Type=Integer,Description="RMS Mapping Quality">',
    '##FORMAT=<ID=QD,Number=1,Type=Integer,Description="Quality By Depth (Variant quality / read depth (non-converted reads only))">',
    '##FORMAT=<ID=MC8,Number=8,Type=Integer,Description="Base counts: non-informative for methylation (ACGT) followed by informative for methylation (ACGT)">',
    '##FORMAT=<ID=AMQ,Number=.,Type=Integer,Description="Average base quailty for where MC8 base count non-zero">',
    '##FORMAT=<ID=CS,Number=1,Type=String,Description="Strand of Cytosine relative to reference sequence (+/-/+-/NA)">',
    '##FORMAT=<ID=CG,Number=1,Type=String,Description="CpG Status (from genotype calls: Y/N/H/?)">',
    '##FORMAT=<ID=CX,Number=1,Type=String,Description="5 base sequence context (from position -2 to +2 on the positive strand) determined from genotype call">',
    '##FORMAT=<ID=FS,Number=1,Type=Integer,Description="Phred scaled log p-value from Fishers exact test of strand bias">',
)


def header_text(contigs, sample, under_conv=0.01, over_conv=0.05, mapq_thresh=20, min_qual=20, date=None, dbsnp_header=None,
                benchmark_mode=False, version="2.1", fileformat="VCFv4.2"):
    """The VCF header print_vcf_header assembles (src/print_vcf.c:621-731): fileformat, htslib's PASS filter, the date /
    source / dbsnp lines (not in benchmark mode), one ##contig per (name, length[, assembly, md5, species]), the INFO /
    FILTER / FORMAT definitions, the column line.  The last FORMAT line is closed here (the reference's text lacks its '>';
    htslib's header parser decides what becomes of it — unpinned).  `fileformat` is htslib's bcf_hdr_get_version()."""
    import time

    lines = ["##fileformat=%s" % fileformat, '##FILTER=<ID=PASS,Description="All filters passed">']
    if not benchmark_mode:
        d = date or (lambda t: (t.tm_mday, t.tm_mon, t.tm_year))(time.localtime())
        lines.append("##fileDate(dd/mm/yyyy)=%02d/%02d/%04d" % tuple(d))
        lines.append("##source=bs_call_v%s,under_conversion=%g,over_conversion=%g,mapq_thresh=%d,bq_thresh=%d"
                     % (version, under_conv, over_conv, mapq_thresh, min_qual))
        if dbsnp_header:
            lines.append("##dbsnp=<%s>" % dbsnp_header)
    for c in contigs:
        extra = "".join(",%s=%s" % (k, v) for k, v in zip(("assembly", "md5", "sp"), c[2:]) if v)
        lines.append("##contig=<ID=%s,length=%d%s>" % (c[0], c[1], extra))
    lines += INFO_FILTER_FORMAT_LINES
    lines.append(HEADER % sample)
    return "\n".join(lines) + "\n"


def bcf_block(recs, rid, dbsnp=None):
    """The BCF2 records of a block's packed records (VCF_REC[]), concatenated (bsc_bcf_block); dbsnp: a DbSnpIndex with the
    block's contig loaded, to name the records whose rs_found flag is set."""
    import ctypes as C

    import numpy as np

    from . import _lib
    from .abi import VCF_REC

    L = _lib.load()
    recs = np.ascontiguousarray(recs, dtype=VCF_REC)
    ids = _lib.BcfIds()
    L.bsc_bcf_default_ids(C.byref(ids))
    cap = 64 + 256 * max(1, len(recs))
    buf = np.empty(cap, dtype=np.uint8)
    done = C.c_uint64(0)
    n = L.bsc_bcf_block(recs.ctypes.data, len(recs), rid, C.byref(ids), None if dbsnp is None else dbsnp._h, buf.ctypes.data, cap,
                        C.byref(done))
    if n < 0 or done.value != len(recs):
        raise RuntimeError("bsc_bcf_block failed (%d, %d of %d records)" % (n, done.value, len(recs)))
    return buf[:n].tobytes()


def _bgzf_block(data: bytes) -> bytes:
    """One BGZF block (SAM specification section 4.1): a gzip member with the 'BC' extra field holding the block size."""
    import struct
    import zlib

    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(data) + co.flush()
    total = 18 + len(body) + 8
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", total - 1) + body
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def write_bcf(path, header: str, record_blocks, compressed=True):
    """A BCF file: magic, header text, the records (an iterable of bsc_bcf_block outputs), BGZF-compressed like the
    reference's default output ("wb") or plain ("wbu")."""
    import struct

    text = header.encode() + b"\0"
    head = b"BCF\x02\x02" + struct.pack("<I", len(text)) + text
    with open(path, "wb") as f:
        if not compressed:
            f.write(head)
            for b in record_blocks:
                f.write(b)
            return
        pend = bytearray(head)
        for b in record_blocks:
            pend += b
            while len(pend) >= 0xFF00:
                f.write(_bgzf_block(bytes(pend[:0xFF00])))
                del pend[:0xFF00]
        if pend:
            f.write(_bgzf_block(bytes(pend)))
        f.write(BGZF_EOF)


def write_vcf(path, header: str, line_blocks, bgzip=False):
    """A VCF text file: the header, then the data lines (an iterable of lists of lines, e.g. format_records_c outputs);
    bgzip=True writes BGZF blocks (the reference's "wz" mode).  The GL floats are printed with %g (see the module text)."""
    def chunks():
        yield header.encode()
        for lines in line_blocks:
            if lines:
                yield ("\n".join(lines) + "\n").encode()

    with open(path, "wb") as f:
        if not bgzip:
            for c in chunks():
                f.write(c)
            return
        pend = bytearray()
        for c in chunks():
            pend += c
            while len(pend) >= 0xFF00:
                f.write(_bgzf_block(bytes(pend[:0xFF00])))
                del pend[:0xFF00]
        if pend:
            f.write(_bgzf_block(bytes(pend)))
        f.write(BGZF_EOF)
