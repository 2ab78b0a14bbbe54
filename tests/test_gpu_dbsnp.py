"""BASELINE.json configs[4] in small: a contig with a dbSNP index loaded.  The index is written in the reference's on-disk
format (tools/make_dbsnp_index.py), read back through the library's C reader (csrc/dbsnp.c), its per-position flags
drive the fused chain on the GPU, and records + statistics must equal the CPU oracle's (calc threads + print thread with
the same flags): the forced AA / TT homozygous-reference records of fq_mask sites (rs_found & 2, src/print_vcf.c:139) and
the dbSNP counters of the statistics (:426-441) included."""
import importlib.util
import os

import numpy as np
import pytest

import bs_call_amd as B
from bs_call_amd.abi import SITE_STATS, VCF_CORE
from bs_call_amd.dbsnp import DbSnpIndex

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("make_dbsnp_index", os.path.join(ROOT, "tools", "make_dbsnp_index.py"))
W = importlib.util.module_from_spec(spec)
spec.loader.exec_module(W)


def test_contig_with_dbsnp_index(tmp_path, oracle, tables, libm_exact):
    import torch

    from tests.test_gpu_chain import _same_core, _same_stats

    n, x0, cov = 1_200_000, 1, 30
    path = str(tmp_path / "cfg4.idx")
    sites = W.synthetic_sites(n, 300)
    W.write_index(path, {"ctgS": sites, "other": W.synthetic_sites(5_000, 300)})
    with DbSnpIndex(path) as db:
        assert db.load_contig("ctgS") == len(sites)
        flags = db.flags(x0, n)
        names = {pos: db.name(pos) for pos, _, _, _ in sites[:50]}
    assert int((flags != 0).sum()) == len(sites) and 0.05 < float((flags == 3).sum()) / len(sites) < 0.16
    pile, ref2 = B.synth_pileup_host(88172645463325252 + 3, 5_000_000, n + 2, cov, 1)
    pile = pile[:n]
    dev = "cuda:0"
    d_cts = torch.from_numpy(pile.view(np.uint8).reshape(-1)).to(dev)
    d_ref = torch.from_numpy(ref2).to(dev)
    d_db = torch.from_numpy(flags).to(dev)
    d_core = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    win = 400_020  # a multiple of the 60-position wave-tile
    with B.SiteCaller() as c:
        for first in range(0, n, win):
            m = min(win, n - first)
            lc, lr = min(2, first), min(4, first)
            c.chain_device(d_cts.data_ptr() + (first - lc) * 104, d_ref.data_ptr() + (first - lr), x0, n, first, m,
                           d_core.data_ptr() + first * 64, d_dbsnp=d_db.data_ptr() + first, with_stats=True)
        torch.cuda.synchronize()
        got = d_core.cpu().numpy().view(VCF_CORE)
        gst = c.site_stats().copy()
    flav = oracle.LIBM if libm_exact else oracle.BSM
    gtm, skip = oracle.call_sites(pile, ref2[:n], tables, flav, -16)
    stats = np.zeros(1, dtype=SITE_STATS)
    carry = np.zeros(2, dtype=np.uint32)
    exp = oracle.vcf_block_stats(gtm, skip, ref2, x0, stats, carry, tables.lfact_store, False, 1, 0xFFFFFFFF, flags)
    _same_core(got, exp, "configs[4] contig")
    _same_stats(gst, stats[0])
    # the index did something: hom-ref AA / TT records exist only where fq_mask forces them, and the dbSNP counters moved
    homref = (got["emit"] == 1) & (((got["gt"] == 0) & (got["ref_code"] == 1)) | ((got["gt"] == 9) & (got["ref_code"] == 4)))
    assert homref.sum() > 50 and (flags[homref] == 3).all()
    assert int(gst["dbSNP_sites"][0]) == int(((flags != 0) & (got["emit"] == 1)).sum()) > 1000
    # names for the ID column come from the same reader
    for pos, (r, nm, ln) in names.items():
        assert r == flags[pos - x0] and nm.startswith("rs") and nm[2:].isdigit()
