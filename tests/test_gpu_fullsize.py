"""Full-size (BASELINE.json configs[1]: 50 M positions at 30x) checks of the HBM-resident path through properties
that do not need a 50 M-site CPU run: idempotence, window independence, counters = a census of the output, and a
random 1 M-site window against the oracle."""
import numpy as np
import pytest

import bs_call_amd as B

pytestmark = pytest.mark.gpu
SEED = 88172645463325252 + 2
N = 50_000_000
COV = 30


def test_full_contig_properties(oracle, tables, libm_exact):
    import torch

    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    with B.SiteCaller() as c:
        d_cts = torch.empty(N * 104, dtype=torch.uint8, device=dev)
        d_ref = torch.empty(N, dtype=torch.uint8, device=dev)
        d_out = torch.empty(N * 200, dtype=torch.uint8, device=dev)
        d_skip = torch.empty(N, dtype=torch.uint8, device=dev)
        c.synth_device(SEED, 0, N, COV, d_cts.data_ptr(), d_ref.data_ptr(), 0, st)
        c.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), N, d_out.data_ptr(), d_skip.data_ptr(), 200, st)
        torch.cuda.synchronize()
        s = c.stats()

        # (1) the counters are a census of the output records
        rec = d_out.view(N, 200)
        skip = d_skip.bool()
        assert s["sites"] == N and s["covered"] == int((~skip).sum())
        mx = rec[:, 192][~skip].long()
        hist = torch.bincount(mx, minlength=10).cpu().tolist()
        assert s["gt_hist"] == hist
        het = torch.tensor(B.GT_HET.astype(np.uint8), device=dev)[mx].sum().item()
        assert s["het_calls"] == het
        # skipped records are all zero; covered records carry n > 0 counts
        assert int(rec[skip].sum()) == 0
        # fisher_strand is non-zero only on heterozygous calls
        fs = rec[:, 176:184].contiguous().view(torch.float64).view(-1)
        hetmask = torch.tensor(B.GT_HET, device=dev)[rec[:, 192].long()] & ~skip
        assert int((fs[~hetmask] != 0).sum()) == 0

        # (2) idempotence
        d_out2 = torch.zeros_like(d_out)
        c.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), N, d_out2.data_ptr(), d_skip.data_ptr(), 200, st)
        torch.cuda.synchronize()
        assert torch.equal(d_out, d_out2)
        del d_out2

        # (3) window independence: sub-windows (ragged starts and lengths) give the same bytes as the full run
        for a, m in ((0, 4_194_304), (12_345_679, 1_000_001), (N - 777, 777), (33_333_333, 63)):
            w_out = torch.zeros(m * 200, dtype=torch.uint8, device=dev)
            w_skip = torch.zeros(m, dtype=torch.uint8, device=dev)
            # pile-up windows must be 16-byte aligned for the device entry: copy odd-offset windows
            w_cts = d_cts[a * 104 : (a + m) * 104].clone()
            w_ref = d_ref[a : a + m].clone()
            c.call_sites_device(w_cts.data_ptr(), w_ref.data_ptr(), m, w_out.data_ptr(), w_skip.data_ptr(), 200, st)
            torch.cuda.synchronize()
            assert torch.equal(w_out, d_out[a * 200 : (a + m) * 200]) and torch.equal(w_skip, d_skip[a : a + m])

        # (4) a 1 M-site window against the oracle, byte for byte
        a, m = 27_182_818, 1_000_000
        pile = d_cts[a * 104 : (a + m) * 104].cpu().numpy().view(B.PILEUP)
        ref = d_ref[a : a + m].cpu().numpy()
        hp, hr = B.synth_pileup_host(SEED, a, m, COV)
        assert pile.tobytes() == hp.tobytes() and (ref == hr).all()  # device generator == host twin
        exp, eskip = oracle.call_sites(pile, ref, tables, oracle.LIBM if libm_exact else oracle.BSM, -8)
        got = d_out[a * 200 : (a + m) * 200].cpu().numpy().view(B.GT_METH)
        assert got.tobytes() == exp.tobytes() and (d_skip[a : a + m].cpu().numpy() == eskip).all()


def test_config4_deep_coverage_window(oracle, tables, libm_exact):
    """BASELINE.json configs[3]: 10 Mb at 200x (|ll - max| >= 512: the exp fall-back; quality sums still exact in f32) —
    full size through properties (census, idempotence, the fused chain's
    records == the unfused chain's), and a 200 k window byte for byte against the oracle."""
    import torch

    n, cov, seed = 10_000_000, 200, SEED + 7
    dev = torch.device("cuda:0")
    with B.SiteCaller() as c:
        d_cts = torch.empty((n + 2) * 104, dtype=torch.uint8, device=dev)
        d_ref = torch.empty(n + 2, dtype=torch.uint8, device=dev)
        d_out = torch.empty(n * 200, dtype=torch.uint8, device=dev)
        d_skip = torch.empty(n, dtype=torch.uint8, device=dev)
        c.synth_device(seed, 0, n + 2, cov, d_cts.data_ptr(), d_ref.data_ptr(), 0, None)
        c.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out.data_ptr(), d_skip.data_ptr(), 200, None)
        torch.cuda.synchronize()
        s = c.stats()
        rec = d_out.view(n, 200)
        skip = d_skip.bool()
        assert s["sites"] == n and s["covered"] == int((~skip).sum())
        mx = rec[:, 192][~skip].long()
        assert s["gt_hist"] == torch.bincount(mx, minlength=10).cpu().tolist()
        het = torch.tensor(B.GT_HET, device=dev)[rec[:, 192].long()] & ~skip
        assert s["het_calls"] == int(het.sum()) > 5_000
        fs = rec[:, 176:184].contiguous().view(torch.float64).view(-1)
        assert int((fs[~het] != 0).sum()) == 0 and int((fs[het] != 0).sum()) > 0.5 * int(het.sum())
        # depth really is deep (the generator draws 200 +- 25 %: the 256-entry log-factorial table still covers the strand
        # margins here; its lgamma branch is exercised at 300x by tests/test_gpu_parity.py::test_synth_parity)
        depth = d_cts.view(torch.int32).view(n + 2, 26)[:n, 16]
        assert int(depth.max()) >= 240 and float(depth.float().mean()) > 190
        d_out2 = torch.zeros_like(d_out)
        c.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out2.data_ptr(), d_skip.data_ptr(), 200, None)
        torch.cuda.synchronize()
        assert torch.equal(d_out, d_out2)
        del d_out2
        # fused chain == unfused chain on the whole 10 M block (records and statistics)
        d_core_u = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        d_core_f = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        c.reset_site_stats()
        c.vcf_records_device(d_out.data_ptr(), 200, d_skip.data_ptr(), d_ref.data_ptr(), n, 1, d_core_u.data_ptr())
        c.vcf_stats_device(d_core_u.data_ptr(), d_out.data_ptr(), 200, n)
        torch.cuda.synchronize()
        st_u = c.site_stats().copy()
        c.reset_site_stats()
        c.chain_device(d_cts.data_ptr(), d_ref.data_ptr(), 1, n, 0, n, d_core_f.data_ptr(), with_stats=True)
        torch.cuda.synchronize()
        st_f = c.site_stats().copy()
        assert torch.equal(d_core_u, d_core_f)
        from tests.test_gpu_chain import _same_stats

        _same_stats(st_f, st_u)
        # a 200 k window against the oracle, byte for byte
        a, m = 3_141_592, 200_000
        pile = d_cts[a * 104 : (a + m) * 104].cpu().numpy().view(B.PILEUP)
        ref = d_ref[a : a + m].cpu().numpy()
        exp, eskip = oracle.call_sites(pile, ref, tables, oracle.LIBM if libm_exact else oracle.BSM, -8)
        got = d_out[a * 200 : (a + m) * 200].cpu().numpy().view(B.GT_METH)
        assert got.tobytes() == exp.tobytes() and (d_skip[a : a + m].cpu().numpy() == eskip).all()


def test_sub_launch_loop(oracle, tables, libm_exact, monkeypatch):
    """A call longer than one launch's cap (2^31 positions: the heterozygous list holds 32-bit indices) is split into
    sub-launches; with the cap lowered through BSC_MAX_LAUNCH_SITES the loop runs on a block that fits a test."""
    monkeypatch.setenv("BSC_MAX_LAUNCH_SITES", "65536")
    n = 300_001  # 4 full sub-launches and a ragged one
    pile, ref = B.synth_pileup_host(SEED + 9, 0, n, 30)
    with B.SiteCaller() as c:
        got, skip = c.call_sites(pile, ref)
        s = c.stats()
    exp, eskip = oracle.call_sites(pile, ref, tables, oracle.LIBM if libm_exact else oracle.BSM, -8)
    assert got.tobytes() == exp.tobytes() and (skip == eskip).all()
    assert s["sites"] == n and s["covered"] == int((eskip == 0).sum()) and s["het_calls"] == int(B.GT_HET[exp["max_gt"]][eskip == 0].sum())


@pytest.mark.parametrize("sites,cov", [(10_000_000, 200), (50_000_000, 30)])
def test_reads_in_full_size(oracle, tables, libm_exact, sites, cov):
    """HOT LOOP A and the reads-in chain at config sizes (BASELINE.json configs[3]: 10 Mb at 200x, the deep read-stack path;
    configs[1]: 50 Mb at 30x) on device-resident L-reads: the pile-up's n summed over the block = a census of the countable
    bases of the reads; a 200 k-position window of the pile-up and of the records = the oracle chain
    (orc_accumulate -> orc_call_sites -> orc_vcf_block); the records of bsc_reads_chain_device = those of
    bsc_accumulate_device -> bsc_chain_device over the whole block, and its counters a census of them."""
    import torch

    from bs_call_amd import reads as R
    from bs_call_amd.abi import VCF_CORE

    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    x, chunk, win = 1000, 1_000_000, 200_000
    tpl, seq, y = R.synth_block(SEED, x, sites, cov, chunk=chunk)
    n = y - x + 1
    n_pad = (n + 63) // 64 * 64
    ref = B.synth_ref_host(SEED, x, n + 2)
    q = seq >> 2
    census = int(((q >= 20) & (q < 63)).sum())  # every byte of seq belongs to one read; every read lies inside the block
    with B.SiteCaller() as c:
        d_tpl = torch.from_numpy(tpl.view(np.uint8).reshape(-1)).to(dev)
        d_seq = torch.from_numpy(seq).to(dev)
        d_ref = torch.from_numpy(ref).to(dev)
        d_pile = torch.zeros((n_pad + 2) * 104, dtype=torch.uint8, device=dev)
        c.accumulate_device(d_tpl.data_ptr(), len(tpl), d_seq.data_ptr(), seq.size, x, y, d_pile.data_ptr(), st)
        c.block_status(st)
        rows = d_pile[: n * 104].view(n, 104).view(torch.int32).view(n, 26)
        assert int(rows[:, 16].sum(dtype=torch.int64)) == census
        assert int(rows[:, :16].sum(dtype=torch.int64)) == census
        # the first chunk's templates alone cover its positions: compare a window with the oracle
        t1, s1, y1 = R.synth_block(SEED, x, min(chunk, sites), cov, chunk=chunk)
        rc, pile = oracle.accumulate(t1, s1, x, y1, 20)
        assert rc == 0 and d_pile[: win * 104].cpu().numpy().tobytes() == pile[:win].tobytes()
        # the reads-in chain against the unfused route on the device, whole block
        d_core = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        d_core2 = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        c.reset_stats()
        c.reads_chain_device(d_tpl.data_ptr(), len(tpl), d_seq.data_ptr(), seq.size, x, y, d_ref.data_ptr(), d_core.data_ptr(), stream=st)
        c.block_status(st)
        cnt = c.stats()
        c.chain_device(d_pile.data_ptr(), d_ref.data_ptr(), x, n, 0, n, d_core2.data_ptr(), stream=st)
        torch.cuda.synchronize()
        assert torch.equal(d_core, d_core2)
        covered = int((rows[:, 16] != 0).sum())
        assert cnt["sites"] == n and cnt["covered"] == covered
        if libm_exact:
            gtm, skip = oracle.call_sites(pile[: win + 8], ref[: win + 8], tables, oracle.LIBM, -8)
            core = oracle.vcf_block(gtm, skip, ref[: win + 10], x)
            assert d_core[: win * 64].cpu().numpy().tobytes() == core[:win].tobytes()
