"""ctypes binding of libbscall_amd.so (include/bscall_amd.h).  No fallback: a missing library is an error."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# BSCALL_AMD_LIB: developer override used to A/B kernel build variants (tools/ab_variants.sh)
LIB_PATH = os.environ.get("BSCALL_AMD_LIB") or os.path.join(HERE, "lib", "libbscall_amd.so")

# every symbol include/bscall_amd.h declares
EXPORTS = (
    "bsc_abi_version",
    "bsc_last_error",
    "bsc_params_default",
    "bsc_create",
    "bsc_destroy",
    "bsc_get_tables",
    "bsc_phred_table",
    "bsc_call_sites",
    "bsc_alloc_host",
    "bsc_free_host",
    "bsc_call_sites_device",
    "bsc_accumulate",
    "bsc_accumulate_device",
    "bsc_block_status",
    "bsc_last_accumulate_ms",
    "bsc_call_block",
    "bsc_block_submit",
    "bsc_block_fetch",
    "bsc_block_submit_to",
    "bsc_synth_reads_host",
    "bsc_vcf_records",
    "bsc_vcf_records_device",
    "bsc_vcf_format",
    "bsc_vcf_format_rec",
    "bsc_vcf_compact_device",
    "bsc_block_records",
    "bsc_block_records_submit",
    "bsc_block_records_submit_inplace",
    "bsc_block_records_fetch",
    "bsc_blocks_records_submit",
    "bsc_blocks_records_submit_inplace",
    "bsc_blocks_records_fetch",
    "bsc_blocks_records",
    "bsc_blocks_submit_to",
    "bsc_blocks_submit_to_inplace",
    "bsc_vcf_stats",
    "bsc_vcf_stats_device",
    "bsc_get_site_stats",
    "bsc_reset_site_stats",
    "bsc_chain_device",
    "bsc_reads_chain_device",
    "bsc_last_reads_chain_ms",
    "bsc_last_chain_ms",
    "bsc_get_site_totals",
    "bsc_gc_bins",
    "bsc_set_gc_bins",
    "bsc_set_gc_bins_host",
    "bsc_get_gc_stats",
    "bsc_report_json",
    "bsc_bcf_default_ids",
    "bsc_bcf_record",
    "bsc_bcf_block",
    "bsc_bcf_block_device",
    "bsc_bcf_sites_device",
    "bsc_block_bcf",
    "bsc_block_bcf_raw",
    "bsc_block_bcf_submit",
    "bsc_block_bcf_submit_inplace",
    "bsc_block_bcf_fetch",
    "bsc_blocks_bcf_submit",
    "bsc_blocks_bcf_submit_inplace",
    "bsc_blocks_bcf_fetch",
    "bsc_dbsnp_names",
    "bsc_fasta_contig",
    "bsc_block_reference",
    "bsc_bam_open",
    "bsc_bam_open_threads",
    "bsc_bam_close",
    "bsc_bam_n_refs",
    "bsc_bam_ref_name",
    "bsc_bam_ref_len",
    "bsc_bam_header_text",
    "bsc_bam_next_block",
    "bsc_bam_filter_counts",
    "bsc_bam_malformed",
    "bsc_chain_window_quantum",
    "bsc_chain_window_size",
    "bsc_prepare_templates",
    "bsc_prepare_templates_profile",
    "bsc_prepare_templates_device",
    "bsc_block_records_raw",
    "bsc_block_start",
    "bsc_template_qual",
    "bsc_template_walk_flags",
    "bsc_dbsnp_open",
    "bsc_dbsnp_close",
    "bsc_dbsnp_n_contigs",
    "bsc_dbsnp_contig_name",
    "bsc_dbsnp_header",
    "bsc_dbsnp_load_contig",
    "bsc_dbsnp_flags",
    "bsc_dbsnp_name",
    "bsc_set_profiling",
    "bsc_set_reads_fused",
    "bsc_block_bcf_again",
    "bsc_bcf_stream_detach",
    "bsc_detached_read",
    "bsc_detached_wait",
    "bsc_detached_free",
    "bsc_debug_fail_summary_alloc",
    "bsc_last_kernel_ms",
    "bsc_kernel_ms_history",
    "bsc_synchronize",
    "bsc_stream_probe_ms",
    "bsc_get_stats",
    "bsc_reset_stats",
    "bsc_synth_pileup_device",
    "bsc_synth_pileup_host",
    "bsc_bamstream_open",
    "bsc_bamstream_open_contigs",
    "bsc_bamdev_open_contigs",
    "bsc_bamstream_close",
    "bsc_bamstream_next",
    "bsc_bamstream_release",
    "bsc_bamstream_n_refs",
    "bsc_bamstream_ref_name",
    "bsc_bamstream_ref_len",
    "bsc_bamstream_header_text",
    "bsc_bamstream_first_record",
    "bsc_bamstream_threads",
    "bsc_bamstream_default_threads",
    "bsc_bamdev_open",
    "bsc_bamdev_close",
    "bsc_bamdev_n_refs",
    "bsc_bamdev_ref_name",
    "bsc_bamdev_ref_len",
    "bsc_bamdev_header_text",
    "bsc_bamdev_next_block",
    "bsc_bamdev_fetch_block",
    "bsc_bamdev_filter_counts",
    "bsc_bamdev_malformed",
    "bsc_bamdev_run_stats",
    "bsc_block_records_rawdev",
    "bsc_block_bcf_rawdev",
    "bsc_block_bcf_rawdev_keep",
    "bsc_bcf_stream_read",
    "bsc_inflate_raw",
    "bsc_crc32",
    "bsc_last_raw_block_ms",
    "bsc_reads_chain_len_device",
    "bsc_bcf_sites_len_device",
)


class Params(C.Structure):
    _fields_ = [
        ("under_conv", C.c_double),
        ("over_conv", C.c_double),
        ("ref_bias", C.c_double),
        ("min_qual", C.c_int32),
        ("device", C.c_int32),
    ]


class VcfParams(C.Structure):
    _fields_ = [("all_positions", C.c_int32), ("reg_start", C.c_uint32), ("reg_stop", C.c_uint32)]


class Window(C.Structure):
    _fields_ = [("x", C.c_uint32), ("n_block", C.c_uint32), ("first", C.c_uint32), ("n", C.c_uint32)]


class ContigTotals(C.Structure):
    _fields_ = [("name", C.c_char_p)] + [(f, C.c_uint64 * 2) for f in
                                         ("snps", "indels", "multi", "dbSNP_sites", "dbSNP_var", "CpG_ref", "CpG_nonref")]


class Report(C.Structure):
    _fields_ = [
        ("under_conv", C.c_double),
        ("over_conv", C.c_double),
        ("mapq_thresh", C.c_int32),
        ("min_qual", C.c_int32),
        ("day", C.c_int32),
        ("month", C.c_int32),
        ("year", C.c_int32),
        ("have_dbsnp", C.c_int32),
        ("filter_cts", C.c_uint64 * 15),
        ("filter_bases", C.c_uint64 * 15),
        ("base_filter", C.c_uint64 * 5),
        ("total", C.c_void_p),
        ("gc", C.c_void_p),
        ("read_profile", C.c_void_p),
        ("n_read_profile", C.c_uint32),
        ("n_contigs", C.c_uint32),
        ("contigs", C.POINTER(ContigTotals)),
    ]


class ReadProfile(C.Structure):
    _fields_ = [("ref", C.c_void_p), ("x", C.c_uint32), ("n_ref", C.c_uint32), ("counts", C.c_void_p), ("cap", C.c_uint32),
                ("used", C.c_uint32)]


class ReaderParams(C.Structure):
    _fields_ = [("mapq_thresh", C.c_uint32), ("max_template_len", C.c_uint64), ("keep_unmatched", C.c_int32),
                ("ignore_duplicates", C.c_int32), ("keep_duplicates", C.c_int32), ("region_tid", C.c_int32), ("region_start", C.c_uint32),
                ("region_stop", C.c_uint32)]


class ReadBlock(C.Structure):
    _fields_ = [("tid", C.c_int32), ("y", C.c_uint32), ("nr", C.c_uint32), ("tpl", C.c_void_p), ("seq", C.c_void_p),
                ("seq_bytes", C.c_uint64), ("misms", C.c_void_p), ("n_misms", C.c_uint64)]


class BamSlab(C.Structure):
    _fields_ = [("bytes", C.c_void_p), ("stream_off", C.c_uint64), ("n_bytes", C.c_uint32), ("rec_off", C.c_void_p), ("n_recs", C.c_uint32),
                ("last", C.c_int32), ("seq", C.c_uint64)]


class DevReadBlock(C.Structure):
    _fields_ = [("tid", C.c_int32), ("x", C.c_uint32), ("y", C.c_uint32), ("nr", C.c_uint32), ("d_tpl", C.c_void_p), ("d_seq", C.c_void_p),
                ("seq_bytes", C.c_uint64), ("d_misms", C.c_void_p), ("n_misms", C.c_uint64), ("ins_pad", C.c_uint64)]


class BcfIds(C.Structure):
    _fields_ = [(f, C.c_int32) for f in ("pass_", "fail", "mac1", "info_cx", "fmt_gt", "fmt_ft", "fmt_gl", "fmt_gq", "fmt_dp",
                                          "fmt_mq", "fmt_qd", "fmt_mc8", "fmt_amq", "fmt_cs", "fmt_cg", "fmt_cx", "fmt_fs")]


class BcfNames(C.Structure):
    """bsc_bcf_names: the dbSNP names of a block's flagged positions (host arrays)."""

    _fields_ = [("pos", C.c_void_p), ("off", C.c_void_p), ("bytes", C.c_void_p), ("n", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [
        ("sites", C.c_uint64),
        ("covered", C.c_uint64),
        ("gt_hist", C.c_uint64 * 10),
        ("het_calls", C.c_uint64),
        ("reserved", C.c_uint64 * 3),
    ]


_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7; if this library
    pulled in /opt/rocm's copy first, a later `import torch` would mix the two runtimes and find no GPU.
    When torch is installed (it owns device memory and streams in bench.py and the tests), load its copy
    first so that libbscall_amd.so's NEEDED libamdhip64.so.7 resolves to it.  Without torch the system
    runtime is used."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return  # torch already loaded its runtime; the dynamic linker will reuse it by SONAME
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


ABI_VERSION = 2  # BSC_ABI_VERSION of include/bscall_amd.h


def load():
    """Load the shared library; raise (never fall back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "bs_call_amd: %s is missing. Build it with `make` at the repository root "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % LIB_PATH
        )
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    if os.environ.get("BSCALL_AMD_LIB"):
        # a developer's A/B build of an older revision may lack newer entries: bind what it has (tools/ab_*.py)
        class _Tolerant:
            def __init__(self, lib):
                object.__setattr__(self, "_lib", lib)

            def __getattr__(self, name):
                try:
                    return getattr(self._lib, name)
                except AttributeError:
                    class _Missing:
                        restype = None
                        argtypes = None

                        def __call__(self, *a):
                            raise ImportError("%s is not in %s" % (name, LIB_PATH))

                    m = _Missing()
                    object.__setattr__(self, name, m)
                    return m

        L = _Tolerant(L)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    L.bsc_abi_version.restype = i32
    L.bsc_abi_version.argtypes = []
    if L.bsc_abi_version() != ABI_VERSION:  # the structs of abi.py mirror ONE header revision
        raise ImportError("%s has ABI %d, this package mirrors ABI %d: rebuild with `make`" % (LIB_PATH, L.bsc_abi_version(), ABI_VERSION))
    L.bsc_last_error.restype = C.c_char_p
    L.bsc_last_error.argtypes = []
    L.bsc_params_default.restype = None
    L.bsc_params_default.argtypes = [C.POINTER(Params)]
    L.bsc_create.restype = i32
    L.bsc_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.bsc_destroy.restype = i32
    L.bsc_destroy.argtypes = [vp]
    L.bsc_get_tables.restype = i32
    L.bsc_get_tables.argtypes = [vp, vp, vp]
    L.bsc_phred_table.restype = i32
    L.bsc_phred_table.argtypes = [vp, vp]
    L.bsc_call_sites.restype = i32
    L.bsc_call_sites.argtypes = [vp, vp, vp, u64, vp, u32, vp]
    L.bsc_alloc_host.restype = vp
    L.bsc_alloc_host.argtypes = [u64]
    L.bsc_free_host.restype = None
    L.bsc_free_host.argtypes = [vp]
    L.bsc_call_sites_device.restype = i32
    L.bsc_call_sites_device.argtypes = [vp, vp, vp, u64, vp, u32, vp, vp]
    L.bsc_accumulate.restype = i32
    L.bsc_accumulate.argtypes = [vp, vp, u32, vp, u64, u32, u32, vp]
    L.bsc_accumulate_device.restype = i32
    L.bsc_accumulate_device.argtypes = [vp, vp, u32, vp, u64, u32, u32, vp, vp]
    L.bsc_block_status.restype = i32
    L.bsc_block_status.argtypes = [vp, vp]
    L.bsc_last_accumulate_ms.restype = i32
    L.bsc_last_accumulate_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.bsc_call_block.restype = i32
    L.bsc_call_block.argtypes = [vp, vp, u32, vp, u64, u32, u32, vp, vp, u32, vp]
    L.bsc_block_submit.restype = i32
    L.bsc_block_submit.argtypes = [vp, vp, u32, vp, u64, u32, u32, vp, u32]
    L.bsc_block_fetch.restype = i32
    L.bsc_block_fetch.argtypes = [vp, vp, vp]
    L.bsc_block_submit_to.restype = i32
    L.bsc_block_submit_to.argtypes = [vp, vp, u32, vp, u64, u32, u32, vp, vp, u32, vp]
    L.bsc_synth_reads_host.restype = C.c_int64
    L.bsc_synth_reads_host.argtypes = [u64, u32, u32, u32, u32, vp, u64, vp, u64, C.POINTER(u64)]
    L.bsc_vcf_records.restype = i32
    L.bsc_vcf_records.argtypes = [vp, vp, u32, vp, vp, vp, u32, u32, C.POINTER(VcfParams), vp]
    L.bsc_vcf_records_device.restype = i32
    L.bsc_vcf_records_device.argtypes = [vp, vp, u32, vp, vp, vp, u32, u32, C.POINTER(VcfParams), vp, vp]
    L.bsc_vcf_format.restype = i32
    L.bsc_vcf_format.argtypes = [vp, vp, C.c_char_p, C.c_char_p, vp, C.c_size_t]
    L.bsc_chain_device.restype = i32
    L.bsc_chain_device.argtypes = [vp, vp, vp, vp, C.POINTER(Window), C.POINTER(VcfParams), i32, vp, vp]
    L.bsc_reads_chain_device.restype = i32
    L.bsc_reads_chain_device.argtypes = [vp, vp, u32, vp, u64, u32, u32, vp, vp, C.POINTER(VcfParams), i32, vp, vp, vp]
    L.bsc_last_reads_chain_ms.restype = i32
    L.bsc_last_reads_chain_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.bsc_gc_bins.restype = i32
    L.bsc_gc_bins.argtypes = [vp, u64, C.POINTER(u32), vp, u64, C.POINTER(u64)]
    L.bsc_set_gc_bins.restype = i32
    L.bsc_set_gc_bins.argtypes = [vp, vp, u32, u32]
    L.bsc_set_gc_bins_host.restype = i32
    L.bsc_set_gc_bins_host.argtypes = [vp, vp, u32, u32]
    L.bsc_get_gc_stats.restype = i32
    L.bsc_get_gc_stats.argtypes = [vp, vp]
    L.bsc_get_site_totals.restype = i32
    L.bsc_get_site_totals.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.bsc_bcf_default_ids.restype = None
    L.bsc_bcf_default_ids.argtypes = [C.POINTER(BcfIds)]
    L.bsc_bcf_record.restype = C.c_long
    L.bsc_bcf_record.argtypes = [vp, i32, C.c_char_p, C.c_size_t, C.POINTER(BcfIds), vp, C.c_size_t]
    L.bsc_fasta_contig.restype = i32
    L.bsc_fasta_contig.argtypes = [C.c_char_p, C.c_char_p, vp, u64, C.POINTER(u64)]
    L.bsc_block_reference.restype = i32
    L.bsc_block_reference.argtypes = [vp, u64, u32, u32, vp]
    L.bsc_bam_open.restype = i32
    L.bsc_bam_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.bsc_bam_open_threads.restype = i32
    L.bsc_bam_open_threads.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    L.bsc_bam_close.restype = None
    L.bsc_bam_close.argtypes = [vp]
    L.bsc_bam_n_refs.restype = i32
    L.bsc_bam_n_refs.argtypes = [vp]
    L.bsc_bam_ref_name.restype = C.c_char_p
    L.bsc_bam_ref_name.argtypes = [vp, i32]
    L.bsc_bam_ref_len.restype = u32
    L.bsc_bam_ref_len.argtypes = [vp, i32]
    L.bsc_bam_header_text.restype = C.c_char_p
    L.bsc_bam_header_text.argtypes = [vp]
    L.bsc_bam_next_block.restype = i32
    L.bsc_bam_next_block.argtypes = [vp, C.POINTER(ReaderParams), C.POINTER(ReadBlock)]
    L.bsc_bam_filter_counts.restype = None
    L.bsc_bam_filter_counts.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.bsc_bam_malformed.restype = C.c_uint64
    L.bsc_bam_malformed.argtypes = [vp]
    L.bsc_bcf_block.restype = C.c_long
    L.bsc_bcf_block.argtypes = [vp, u64, i32, C.POINTER(BcfIds), vp, vp, C.c_size_t, C.POINTER(u64)]
    L.bsc_report_json.restype = C.c_long
    L.bsc_report_json.argtypes = [C.POINTER(Report), C.c_char_p, C.c_size_t]
    L.bsc_chain_window_size.restype = C.c_uint32
    L.bsc_chain_window_size.argtypes = [vp, C.c_uint32]
    L.bsc_chain_window_quantum.restype = C.c_uint32
    L.bsc_chain_window_quantum.argtypes = [vp]
    L.bsc_last_chain_ms.restype = i32
    L.bsc_last_chain_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.bsc_prepare_templates.restype = i32
    L.bsc_prepare_templates.argtypes = [vp, u32, vp, u64, vp, u64, vp, vp, vp, u64, C.POINTER(u64), vp]
    L.bsc_prepare_templates_profile.restype = i32
    L.bsc_prepare_templates_profile.argtypes = [vp, u32, vp, u64, vp, u64, vp, vp, vp, u64, C.POINTER(u64), vp, C.POINTER(ReadProfile)]
    L.bsc_block_start.restype = u32
    L.bsc_block_start.argtypes = [vp]
    L.bsc_template_qual.restype = u32
    L.bsc_template_qual.argtypes = [vp, vp]
    L.bsc_prepare_templates_device.restype = i32
    L.bsc_prepare_templates_device.argtypes = [vp, vp, u32, vp, u64, vp, u64, vp, vp, vp, u64, vp, vp, vp, vp]
    L.bsc_block_records_raw.restype = i32
    L.bsc_block_records_raw.argtypes = [vp, vp, u32, vp, u64, vp, u64, vp, u32, u32, vp, vp, vp, i32, vp, u64, vp, vp, vp]
    L.bsc_block_bcf.restype = i32
    L.bsc_block_bcf.argtypes = [vp, vp, u32, vp, u64, u32, u32, vp, vp, vp, i32, i32, C.POINTER(BcfIds), vp, vp, u64, C.POINTER(u64), C.POINTER(u64)]
    L.bsc_block_bcf_raw.restype = i32
    L.bsc_block_bcf_raw.argtypes = [vp, vp, u32, vp, u64, vp, u64, vp, u32, u32, vp, vp, vp, i32, i32, C.POINTER(BcfIds), vp, vp, u64,
                                    C.POINTER(u64), C.POINTER(u64), vp, vp]
    L.bsc_block_bcf_submit.restype = i32
    L.bsc_block_bcf_submit.argtypes = [vp, vp, u32, vp, u64, u32, u32, vp, vp, vp, i32, i32, C.POINTER(BcfIds), vp, vp, u64]
    L.bsc_block_bcf_submit_inplace.restype = i32
    L.bsc_block_bcf_submit_inplace.argtypes = L.bsc_block_bcf_submit.argtypes
    L.bsc_block_bcf_fetch.restype = i32
    L.bsc_block_bcf_fetch.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.bsc_blocks_bcf_submit.restype = i32
    L.bsc_blocks_bcf_submit.argtypes = [vp, vp, u32, vp, vp, u64, vp, vp, vp, i32, i32, C.POINTER(BcfIds), vp, vp, u64]
    L.bsc_blocks_bcf_submit_inplace.restype = i32
    L.bsc_blocks_bcf_submit_inplace.argtypes = L.bsc_blocks_bcf_submit.argtypes
    L.bsc_blocks_bcf_fetch.restype = i32
    L.bsc_blocks_bcf_fetch.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.bsc_bcf_block_device.restype = i32
    L.bsc_bcf_block_device.argtypes = [vp, vp, vp, u64, i32, C.POINTER(BcfIds), vp, vp, u64, vp, vp]
    L.bsc_bcf_sites_device.restype = i32
    L.bsc_bcf_sites_device.argtypes = [vp, vp, vp, u32, i32, C.POINTER(BcfIds), vp, vp, u64, vp, vp]
    L.bsc_dbsnp_names.restype = i32
    L.bsc_dbsnp_names.argtypes = [vp, u32, u32, vp, vp, vp, u32, u64, C.POINTER(u32), C.POINTER(u64)]
    L.bsc_template_walk_flags.restype = u32
    L.bsc_template_walk_flags.argtypes = [vp, u32]
    L.bsc_dbsnp_open.restype = i32
    L.bsc_dbsnp_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.bsc_dbsnp_close.restype = None
    L.bsc_dbsnp_close.argtypes = [vp]
    L.bsc_dbsnp_n_contigs.restype = i32
    L.bsc_dbsnp_n_contigs.argtypes = [vp]
    L.bsc_dbsnp_contig_name.restype = C.c_char_p
    L.bsc_dbsnp_contig_name.argtypes = [vp, i32]
    L.bsc_dbsnp_header.restype = C.c_char_p
    L.bsc_dbsnp_header.argtypes = [vp]
    L.bsc_dbsnp_load_contig.restype = i32
    L.bsc_dbsnp_load_contig.argtypes = [vp, C.c_char_p, C.POINTER(u64)]
    L.bsc_dbsnp_flags.restype = i32
    L.bsc_dbsnp_flags.argtypes = [vp, u32, u32, vp]
    L.bsc_dbsnp_name.restype = i32
    L.bsc_dbsnp_name.argtypes = [vp, u32, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.bsc_set_reads_fused.restype = i32
    L.bsc_set_reads_fused.argtypes = [vp, i32]
    L.bsc_bcf_stream_detach.restype = i32
    L.bsc_bcf_stream_detach.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.bsc_detached_read.restype = i32
    L.bsc_detached_read.argtypes = [vp, vp, u64, u64, vp]
    L.bsc_detached_wait.restype = i32
    L.bsc_detached_wait.argtypes = [vp]
    L.bsc_detached_free.restype = i32
    L.bsc_detached_free.argtypes = [vp, vp]
    L.bsc_block_bcf_again.restype = i32
    L.bsc_block_bcf_again.argtypes = [vp, vp, u64, C.POINTER(u64), C.POINTER(u64)]
    L.bsc_debug_fail_summary_alloc.restype = i32
    L.bsc_debug_fail_summary_alloc.argtypes = [vp, i32]
    L.bsc_set_profiling.restype = i32
    L.bsc_set_profiling.argtypes = [vp, i32]
    L.bsc_last_kernel_ms.restype = i32
    L.bsc_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.bsc_kernel_ms_history.restype = i32
    L.bsc_kernel_ms_history.argtypes = [vp, u32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.bsc_synchronize.restype = i32
    L.bsc_synchronize.argtypes = [vp]
    L.bsc_get_stats.restype = i32
    L.bsc_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.bsc_reset_stats.restype = i32
    L.bsc_reset_stats.argtypes = [vp]
    L.bsc_stream_probe_ms.restype = i32
    L.bsc_stream_probe_ms.argtypes = [vp, vp, vp, u64, vp, vp, i32, vp, C.POINTER(C.c_float)]
    L.bsc_vcf_compact_device.restype = i32
    L.bsc_vcf_compact_device.argtypes = [vp, vp, vp, u32, vp, u32, vp, u64, vp, vp]
    L.bsc_block_records.restype = i32
    L.bsc_block_records.argtypes = [vp, vp, u32, vp, u64, u32, u32, vp, vp, C.POINTER(VcfParams), i32, vp, u64,
                                    C.POINTER(C.c_uint64)]
    L.bsc_block_records_submit.restype = i32
    L.bsc_block_records_submit.argtypes = [vp, vp, u32, vp, u64, u32, u32, vp, vp, C.POINTER(VcfParams), i32, vp, u64]
    L.bsc_block_records_submit_inplace.restype = i32
    L.bsc_block_records_submit_inplace.argtypes = L.bsc_block_records_submit.argtypes
    L.bsc_block_records_fetch.restype = i32
    L.bsc_block_records_fetch.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.bsc_blocks_records_submit.restype = i32
    L.bsc_blocks_records_submit.argtypes = [vp, vp, u32, vp, vp, u64, vp, vp, vp, i32, vp, u64]
    L.bsc_blocks_records_submit_inplace.restype = i32
    L.bsc_blocks_records_submit_inplace.argtypes = [vp, vp, u32, vp, vp, u64, vp, vp, vp, i32, vp, u64]
    L.bsc_blocks_records_fetch.restype = i32
    L.bsc_blocks_records_fetch.argtypes = [vp, C.POINTER(C.c_uint64), vp]
    L.bsc_blocks_submit_to.restype = i32
    L.bsc_blocks_submit_to.argtypes = [vp, vp, u32, vp, vp, u64, vp, vp, u32, vp, vp]
    L.bsc_blocks_submit_to_inplace.restype = i32
    L.bsc_blocks_submit_to_inplace.argtypes = [vp, vp, u32, vp, vp, u64, vp, vp, u32, vp, vp]
    L.bsc_blocks_records.restype = i32
    L.bsc_blocks_records.argtypes = [vp, vp, u32, vp, vp, u64, vp, vp, vp, i32, vp, u64, C.POINTER(C.c_uint64), vp]
    L.bsc_vcf_format_rec.restype = i32
    L.bsc_vcf_format_rec.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.bsc_vcf_stats_device.restype = i32
    L.bsc_vcf_stats_device.argtypes = [vp, vp, vp, u32, vp, u32, vp]
    L.bsc_vcf_stats.restype = i32
    L.bsc_vcf_stats.argtypes = [vp, vp, vp, u32, vp, u32]
    L.bsc_get_site_stats.restype = i32
    L.bsc_get_site_stats.argtypes = [vp, vp]
    L.bsc_reset_site_stats.restype = i32
    L.bsc_reset_site_stats.argtypes = [vp]
    L.bsc_synth_pileup_device.restype = i32
    L.bsc_synth_pileup_device.argtypes = [vp, u64, u64, u64, u32, u32, vp, vp, vp]
    L.bsc_synth_pileup_host.restype = i32
    L.bsc_synth_pileup_host.argtypes = [u64, u64, u64, u32, u32, vp, vp]
    L.bsc_bamstream_open.restype = i32
    L.bsc_bamstream_open.argtypes = [C.c_char_p, i32, u64, i32, C.POINTER(vp)]
    L.bsc_bamstream_open_contigs.restype = i32
    L.bsc_bamstream_open_contigs.argtypes = [C.c_char_p, i32, u64, i32, vp, i32, C.POINTER(vp)]
    L.bsc_bamdev_open_contigs.restype = i32
    L.bsc_bamdev_open_contigs.argtypes = [vp, C.c_char_p, i32, vp, i32, C.POINTER(vp)]
    L.bsc_bamstream_close.restype = None
    L.bsc_bamstream_close.argtypes = [vp]
    L.bsc_bamstream_next.restype = i32
    L.bsc_bamstream_next.argtypes = [vp, C.POINTER(BamSlab)]
    L.bsc_bamstream_release.restype = i32
    L.bsc_bamstream_release.argtypes = [vp, C.POINTER(BamSlab)]
    L.bsc_bamstream_n_refs.restype = i32
    L.bsc_bamstream_n_refs.argtypes = [vp]
    L.bsc_bamstream_ref_name.restype = C.c_char_p
    L.bsc_bamstream_ref_name.argtypes = [vp, i32]
    L.bsc_bamstream_ref_len.restype = u32
    L.bsc_bamstream_ref_len.argtypes = [vp, i32]
    L.bsc_bamstream_header_text.restype = C.c_char_p
    L.bsc_bamstream_header_text.argtypes = [vp]
    L.bsc_bamstream_first_record.restype = u64
    L.bsc_bamstream_first_record.argtypes = [vp]
    L.bsc_bamstream_threads.restype = i32
    L.bsc_bamstream_threads.argtypes = [vp]
    L.bsc_bamstream_default_threads.restype = i32
    L.bsc_bamstream_default_threads.argtypes = []
    L.bsc_bamdev_open.restype = i32
    L.bsc_bamdev_open.argtypes = [vp, C.c_char_p, i32, C.POINTER(vp)]
    L.bsc_bamdev_close.restype = None
    L.bsc_bamdev_close.argtypes = [vp]
    L.bsc_bamdev_n_refs.restype = i32
    L.bsc_bamdev_n_refs.argtypes = [vp]
    L.bsc_bamdev_ref_name.restype = C.c_char_p
    L.bsc_bamdev_ref_name.argtypes = [vp, i32]
    L.bsc_bamdev_ref_len.restype = u32
    L.bsc_bamdev_ref_len.argtypes = [vp, i32]
    L.bsc_bamdev_header_text.restype = C.c_char_p
    L.bsc_bamdev_header_text.argtypes = [vp]
    L.bsc_bamdev_next_block.restype = i32
    L.bsc_bamdev_next_block.argtypes = [vp, C.POINTER(ReaderParams), C.POINTER(DevReadBlock)]
    L.bsc_bamdev_fetch_block.restype = i32
    L.bsc_bamdev_fetch_block.argtypes = [vp, C.POINTER(DevReadBlock), vp, vp, vp]
    L.bsc_bamdev_filter_counts.restype = i32
    L.bsc_bamdev_filter_counts.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.bsc_bamdev_malformed.restype = u64
    L.bsc_bamdev_malformed.argtypes = [vp]
    L.bsc_bamdev_run_stats.restype = None
    L.bsc_bamdev_run_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(C.c_double)]
    L.bsc_block_records_rawdev.restype = i32
    L.bsc_block_records_rawdev.argtypes = [vp, vp, u32, vp, u64, vp, u64, u64, vp, u32, u32, vp, vp, vp, i32, vp, u64, vp, vp, vp]
    L.bsc_block_bcf_rawdev.restype = i32
    L.bsc_block_bcf_rawdev.argtypes = [vp, vp, u32, vp, u64, vp, u64, u64, vp, u32, u32, vp, vp, vp, i32, i32, C.POINTER(BcfIds), vp, vp, u64,
                                       C.POINTER(u64), C.POINTER(u64), vp, vp]
    L.bsc_block_bcf_rawdev_keep.restype = i32
    L.bsc_block_bcf_rawdev_keep.argtypes = [vp, vp, u32, vp, u64, vp, u64, u64, vp, u32, u32, vp, vp, vp, i32, i32, C.POINTER(BcfIds), vp, u64,
                                            C.POINTER(u64), C.POINTER(u64), vp, vp]
    L.bsc_reads_chain_len_device.restype = i32
    L.bsc_reads_chain_len_device.argtypes = [vp, vp, u32, vp, u64, u32, u32, vp, vp, C.POINTER(VcfParams), i32, vp, vp, vp, vp]
    L.bsc_bcf_sites_len_device.restype = i32
    L.bsc_bcf_sites_len_device.argtypes = [vp, vp, vp, vp, u32, i32, C.POINTER(BcfIds), vp, vp, u64, vp, vp]
    L.bsc_inflate_raw.restype = i32
    L.bsc_inflate_raw.argtypes = [vp, C.c_size_t, vp, C.c_size_t]
    L.bsc_crc32.restype = u32
    L.bsc_crc32.argtypes = [vp, C.c_size_t]
    L.bsc_last_raw_block_ms.restype = i32
    L.bsc_last_raw_block_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.bsc_bcf_stream_read.restype = i32
    L.bsc_bcf_stream_read.argtypes = [vp, u64, u64, vp]
    _lib = L
    return L
