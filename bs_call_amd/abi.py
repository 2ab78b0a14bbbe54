"""Binary layouts shared with the C ABI (include/bscall_amd.h).

These numpy dtypes are byte-for-byte the reference's structs, so arrays of them can be handed to the
library (or memcpy'd to the device) without conversion:
  PILEUP   = `pileup`   include/bs_call.h:174-182   (104 B)
  GT_METH  = `gt_meth`  include/bs_call.h:152-160   (200 B)
  TEMPLATE = one read pair flattened out of `align_details` (include/bs_call.h:64-73), 40 B
Every byte of a record belongs to a named field (`_pad` for the C structs' tail padding): numpy copies records field
by field and would leave unnamed bytes undefined, which breaks byte-wise comparisons of copies.
"""
import numpy as np

PILEUP = np.dtype(
    {
        "names": ["counts", "n", "quality", "mapq2"],
        "formats": [("<u4", (2, 8)), "<u4", ("<f4", (8,)), "<f4"],
        "offsets": [0, 64, 68, 100],
        "itemsize": 104,
    }
)

GT_METH = np.dtype(
    {
        "names": ["counts", "qual", "gt_prob", "fisher_strand", "mq", "aq", "max_gt", "_pad"],
        "formats": [("<u8", (8,)), ("<i4", (8,)), ("<f8", (10,)), "<f8", "<i4", "<i4", "u1", ("u1", (7,))],
        "offsets": [0, 64, 96, 176, 184, 188, 192, 193],
        "itemsize": 200,
    }
)

TEMPLATE = np.dtype(
    {
        "names": ["pos", "len", "off", "mapq", "orientation", "bs_strand", "flags"],
        "formats": [("<u4", (2,)), ("<u4", (2,)), ("<u8", (2,)), ("u1", (2,)), "u1", "u1", "<u4"],
        "offsets": [0, 8, 16, 32, 34, 35, 36],
        "itemsize": 40,
    }
)

# genotype order of gt_prob[] (src/genotype_model.c:110-121)
GENOTYPES = ("AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT")
# heterozygous genotypes (src/init_param.c:16)
GT_HET = np.array([0, 1, 1, 1, 0, 1, 1, 0, 1, 0], dtype=bool)

MAX_QUAL = 43
FLT_QUAL = 63
DEFAULT_UNDER_CONVERSION = 0.01
DEFAULT_OVER_CONVERSION = 0.05
DEFAULT_REF_BIAS = 2.0
DEFAULT_MIN_QUAL = 20

# strand -> pile-up class (src/call_genotypes.c:17-19), 0-based
BASE_TAB_ST = np.array([[0, 1, 2, 3], [0, 5, 2, 7], [4, 1, 6, 3]], dtype=np.int8)

# bsc_vcf_core (include/bscall_amd.h): what the printer derives per position (src/print_vcf.c:32-381), 64 B
VCF_CORE = np.dtype(
    {
        "names": ["pos", "emit", "gt", "ref_code", "gt_enc", "flt", "phred", "n_gl", "cg", "alt", "cx_ref", "cx_gt",
                  "fs", "qd", "dp", "gl", "_pad"],
        "formats": ["<u4", "u1", "u1", "u1", "u1", "u1", "u1", "u1", "S1", "S2", "S5", "S5", "<i4", "<u4", "<u4",
                    ("<f4", (6,)), "<u4"],
        "offsets": [0, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 19, 24, 28, 32, 36, 60],
        "itemsize": 64,
    }
)

# bsc_vcf_rec (include/bscall_amd.h): a written record, packed — the core record + the gt_meth fields the encoder reads
VCF_REC = np.dtype(
    {
        "names": ["core", "counts", "qual", "mq", "aq", "max_gt", "rs_found", "_pad"],
        "formats": [VCF_CORE, ("<u4", (8,)), ("u1", (8,)), "<i4", "<i4", "u1", "u1", ("u1", (14,))],
        "offsets": [0, 64, 96, 104, 108, 112, 113, 114],
        "itemsize": 128,
    }
)

# bsc_site_stats (include/bscall_amd.h): the sum fields of the reference's bs_stats (src/print_vcf.c:382-526)
COV_CAP = 4096
SITE_STATS = np.dtype(
    [
        ("snps", "<u8", (2,)), ("indels", "<u8", (2,)), ("multi", "<u8", (2,)), ("dbSNP_sites", "<u8", (2,)),
        ("dbSNP_var", "<u8", (2,)), ("CpG_ref", "<u8", (2,)), ("CpG_nonref", "<u8", (2,)),
        ("mut_counts", "<u8", (12, 2)), ("dbSNP_mut_counts", "<u8", (12, 2)),
        ("qual", "<u8", (4, 256)),
        ("filter_counts", "<u8", (2, 32)),
        ("qd_stats", "<u8", (256, 2)), ("fs_stats", "<u8", (256, 2)), ("mq_stats", "<u8", (256, 2)),
        ("cov", "<u8", (COV_CAP, 6)),
        ("CpG_ref_meth", "<f8", (2, 101)), ("CpG_nonref_meth", "<f8", (2, 101)),
    ]
)
SITE_STATS_INT_WORDS = (SITE_STATS.itemsize - 4 * 101 * 8) // 8  # the leading u64 part; the rest is 404 doubles

# bsc_raw_template / bsc_misms / bsc_prep_params / bsc_prep_stats (include/bscall_amd.h): read pre-processing
BLOCK_DESC = np.dtype([("x", "<u4"), ("y", "<u4"), ("nr", "<u4"), ("_pad", "<u4")])  # bsc_block_desc (bsc_blocks_records)

RAW_TEMPLATE = np.dtype(
    {
        "names": ["pos", "reference_span", "len", "n_misms", "off", "misms_off", "mapq", "orientation", "bs_strand", "_pad"],
        "formats": [("<u4", (2,)), ("<u4", (2,)), ("<u4", (2,)), ("<u4", (2,)), ("<u8", (2,)), ("<u8", (2,)), ("u1", (2,)), "u1", "u1", "<u4"],
        "offsets": [0, 8, 16, 24, 32, 48, 64, 66, 67, 68],
        "itemsize": 72,
    }
)
MISMS = np.dtype([("type", "<u4"), ("position", "<u4"), ("size", "<u4")])
MISMS_MISMS, MISMS_INS, MISMS_DEL, MISMS_SOFT = 0, 1, 2, 3
PREP_PARAMS = np.dtype([("left_trim", "<i4", (2,)), ("right_trim", "<i4", (2,)), ("min_qual", "<i4")])
PREP_STATS = np.dtype([("base_none", "<u8"), ("base_trim", "<u8"), ("base_clip", "<u8"), ("base_overlap", "<u8"), ("base_lowqual", "<u8"),
                       ("reads", "<u8"), ("read_bases", "<u8")])
