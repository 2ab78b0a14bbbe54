"""bs_call_amd — MI355X-native per-site genotype + methylation caller (bs_call calc-path drop-in).

The compute lives in bs_call_amd/lib/libbscall_amd.so (gfx950 HIP kernels behind the C ABI of
include/bscall_amd.h).  This package is the thin Python host mirror used by tests and bench.py.
"""
from .abi import GENOTYPES, GT_HET, GT_METH, PILEUP, SITE_STATS, SITE_STATS_INT_WORDS, TEMPLATE, VCF_CORE, VCF_REC  # noqa: F401
from .caller import (BscError, BscInexactWarning, PinnedBuffer, SiteCaller, synth_pileup_host, synth_reads_host,  # noqa: F401
                     synth_ref_host)

__all__ = ["SiteCaller", "PinnedBuffer", "BscError", "BscInexactWarning", "synth_pileup_host", "synth_reads_host", "synth_ref_host", "PILEUP", "GT_METH", "TEMPLATE", "GENOTYPES", "GT_HET"]
