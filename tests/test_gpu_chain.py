"""The fused chain (bsc_chain_device: pile-up -> call -> VCF record -> statistics in one pass, csrc/fused.hip) against the
unfused chain (bsc_call_sites_device -> bsc_vcf_records_device -> bsc_vcf_stats_device) and against the CPU oracle
(orc_call_sites + orc_vcf_block_stats = the reference's calc threads followed by its print thread): every byte of every
bsc_vcf_core record, every counter, every integer of the statistics; the methylation profiles to 1e-12 (sum order)."""
import numpy as np
import pytest
import torch

import bs_call_amd as B
from bs_call_amd.abi import SITE_STATS, SITE_STATS_INT_WORDS, VCF_CORE

pytestmark = pytest.mark.gpu
SEED = 88172645463325252
DEV = "cuda:0"


@pytest.fixture(scope="module")
def caller():
    c = B.SiteCaller()
    yield c
    c.close()


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(DEV)


def _unfused(c, pile, ref2, x, dbsnp=None, all_positions=False, reg=(1, 0xFFFFFFFF)):
    """Whole block through the three unfused kernels: (VCF_CORE[n], SITE_STATS, stats dict)."""
    n = len(pile)
    d_cts, d_ref = _dev(pile), _dev(ref2)
    d_out = torch.empty(n * 200, dtype=torch.uint8, device=DEV)
    d_skip = torch.empty(n, dtype=torch.uint8, device=DEV)
    d_core = torch.empty(n * 64, dtype=torch.uint8, device=DEV)
    d_db = None if dbsnp is None else _dev(dbsnp)
    c.reset_stats()
    c.reset_site_stats()
    c.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out.data_ptr(), d_skip.data_ptr(), 200, None)
    c.vcf_records_device(d_out.data_ptr(), 200, d_skip.data_ptr(), d_ref.data_ptr(), n, x, d_core.data_ptr(), all_positions,
                         reg[0], reg[1], None if d_db is None else d_db.data_ptr())
    c.vcf_stats_device(d_core.data_ptr(), d_out.data_ptr(), 200, n, None if d_db is None else d_db.data_ptr())
    torch.cuda.synchronize()
    return d_core.cpu().numpy().view(VCF_CORE).copy(), c.site_stats().copy(), c.stats()


def _fused(c, pile, ref2, x, windows, dbsnp=None, all_positions=False, reg=(1, 0xFFFFFFFF), with_stats=True):
    """The block as a sequence of windows [(first, n), ...] through bsc_chain_device, each window from its OWN buffers
    (only the context the ABI asks for is copied)."""
    nb = len(pile)
    core = np.zeros(nb, dtype=VCF_CORE)
    c.reset_stats()
    c.reset_site_stats()
    for first, n in windows:
        lc, rc, lr = min(2, first), min(2, nb - first - n), min(4, first)
        d_cts = _dev(pile[first - lc : first + n + rc])
        d_ref = _dev(ref2[first - lr : first + n + 2])
        d_db = None if dbsnp is None else _dev(dbsnp[first : first + n])
        d_core = torch.full((n * 64,), 0xA5, dtype=torch.uint8, device=DEV)
        c.chain_device(d_cts.data_ptr(), d_ref.data_ptr(), x, nb, first, n, d_core.data_ptr(), all_positions, reg[0], reg[1],
                       None if d_db is None else d_db.data_ptr(), with_stats, None)
        torch.cuda.synchronize()
        core[first : first + n] = d_core.cpu().numpy().view(VCF_CORE)
    return core, c.site_stats().copy(), c.stats()


def _same_stats(a, b):
    ia = np.frombuffer(a.tobytes(), dtype=np.uint64)[:SITE_STATS_INT_WORDS]
    ib = np.frombuffer(b.tobytes(), dtype=np.uint64)[:SITE_STATS_INT_WORDS]
    if not (ia == ib).all():
        for f in SITE_STATS.names:
            if a[f].dtype.kind == "u":
                assert (a[f] == b[f]).all(), "statistics field %s differs: %s" % (f, np.argwhere(a[f] != b[f])[:5].tolist())
    for f in ("CpG_ref_meth", "CpG_nonref_meth"):
        np.testing.assert_allclose(a[f], b[f], rtol=1e-12, atol=1e-12)


def _same_core(a, b, what):
    if a.tobytes() != b.tobytes():
        for f in VCF_CORE.names:
            x, y = a[f], b[f]
            same = x.tobytes() == y.tobytes()
            if not same:
                idx = [i for i in range(len(a)) if a[i : i + 1][f].tobytes() != b[i : i + 1][f].tobytes()][:5]
                raise AssertionError("%s: field %s differs at %s: %s vs %s" % (what, f, idx, a[f][idx], b[f][idx]))
        raise AssertionError(what + ": padding differs")


def _block(seed, n, cov, flags=0, x0=4000):
    pile, ref = B.synth_pileup_host(seed, x0, n + 2, cov, flags)
    return pile[:n], ref  # ref has n + 2 codes (x .. y + 2)


@pytest.mark.parametrize("n,cov", [(50_000, 30), (61, 30), (62, 30), (121, 10), (1, 30), (2, 30), (3, 30), (59, 200), (60, 30)])
def test_fused_equals_unfused_whole_block(caller, n, cov):
    pile, ref2 = _block(SEED + n, n, cov)
    exp, est, ecnt = _unfused(caller, pile, ref2, 4000)
    got, gst, gcnt = _fused(caller, pile, ref2, 4000, [(0, n)])
    _same_core(got, exp, "whole block n=%d" % n)
    _same_stats(gst, est)
    assert gcnt == ecnt


@pytest.mark.parametrize("cov", [200, 1400])
def test_fused_methylation_beyond_the_lds_pair_table(caller, cov):
    """CpG cytosines with >= 32 informative reads of one kind are counted in the context's 512 x 512 pair table in HBM
    (200x), those with >= 512 are listed and evaluated one by one when the statistics are read (1400x): same profiles as
    the unfused statistics kernel, which evaluates each such cytosine where it meets it."""
    n = 40_000
    pile, ref2 = _block(SEED + 11 + cov, n, cov)
    inf = pile["counts"].sum(axis=1)[:, 4:]
    assert inf.max() >= (512 if cov > 1000 else 64)
    exp, est, ecnt = _unfused(caller, pile, ref2, 4000)
    got, gst, gcnt = _fused(caller, pile, ref2, 4000, [(0, 17_000), (17_000, 23_000)])
    _same_core(got, exp, "deep block")
    _same_stats(gst, est)
    assert gcnt == ecnt
    assert float(gst["CpG_ref_meth"][0].sum() + gst["CpG_nonref_meth"][0].sum()) > 100  # CpGs were profiled
    # reading the statistics consumed the pair table and the list: a second read adds nothing
    again = caller.site_stats().copy()
    _same_stats(again, gst)


def test_fused_windows_equal_whole_block(caller):
    """Windows of every alignment (odd starts take the guarded kernel), a 1-position window, windows that end 0, 1, 2
    positions before the block end: same records and statistics as the block in one piece."""
    n = 40_000
    pile, ref2 = _block(SEED + 5, n, 30, flags=1)
    # make the block end interesting for the flush quirk, and put an N near a window boundary
    ref2 = ref2.copy()
    ref2[9_998] = 0
    exp, est, ecnt = _unfused(caller, pile, ref2, 777)
    cuts = [0, 1, 2, 3, 64, 4_097, 9_999, 10_000, 10_001, 10_062, 25_000, n - 2, n - 1, n]
    windows = [(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])]
    got, gst, gcnt = _fused(caller, pile, ref2, 777, windows)
    _same_core(got, exp, "windowed")
    _same_stats(gst, est)
    assert gcnt == ecnt


def test_fused_vs_oracle_with_dbsnp_and_region(caller, oracle, tables, libm_exact):
    """Against the CPU oracle (calc threads + print thread restated), with dbSNP flags (forced hom-ref emission, dbSNP
    statistics), a region clip and -A."""
    n = 30_000
    pile, ref2 = _block(SEED + 9, n, 30)
    rng = np.random.default_rng(3)
    db = np.zeros(n, dtype=np.uint8)
    idx = rng.choice(n, n // 300, replace=False)
    db[idx] = np.where(rng.random(len(idx)) < 0.1, 3, 1)
    flav = oracle.LIBM if libm_exact else oracle.BSM
    gtm, skip = oracle.call_sites(pile, ref2[:n], tables, flav, -8)
    for allp, reg in ((False, (1, 0xFFFFFFFF)), (True, (1, 0xFFFFFFFF)), (False, (5_000, 20_000))):
        stats = np.zeros(1, dtype=SITE_STATS)
        carry = np.zeros(2, dtype=np.uint32)
        exp = oracle.vcf_block_stats(gtm, skip, ref2, 1000, stats, carry, tables.lfact_store, allp, reg[0], reg[1], db)
        got, gst, _ = _fused(caller, pile, ref2, 1000, [(0, 12_345), (12_345, n - 12_345)], db, allp, reg)
        _same_core(got, exp, "oracle all_positions=%s reg=%s" % (allp, reg))
        _same_stats(gst, stats[0])


def test_fused_deep_coverage_overflow_list(caller):
    """300x: the exp / lgamma fall-backs, and CpG cytosines beyond the 64 x 64 pair table (listed and evaluated)."""
    n = 20_000
    pile, ref2 = _block(SEED + 11, n, 300)
    exp, est, ecnt = _unfused(caller, pile, ref2, 50)
    got, gst, gcnt = _fused(caller, pile, ref2, 50, [(0, 7_000), (7_000, 13_000)])
    _same_core(got, exp, "300x")
    _same_stats(gst, est)
    assert gcnt == ecnt


def test_fused_consecutive_blocks_carry(caller):
    """Two blocks one after the other (the printer's pending cytosine survives from block to block)."""
    n = 5_000
    pile, ref2 = _block(SEED + 13, 2 * n, 30)
    # unfused: block A then block B
    caller.reset_site_stats()
    cores = []
    for k in range(2):
        d_cts, d_ref = _dev(pile[k * n : (k + 1) * n]), _dev(ref2[k * n : (k + 1) * n + 2])
        d_out = torch.empty(n * 200, dtype=torch.uint8, device=DEV)
        d_skip = torch.empty(n, dtype=torch.uint8, device=DEV)
        d_core = torch.empty(n * 64, dtype=torch.uint8, device=DEV)
        caller.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out.data_ptr(), d_skip.data_ptr(), 200, None)
        caller.vcf_records_device(d_out.data_ptr(), 200, d_skip.data_ptr(), d_ref.data_ptr(), n, 100 + k * n, d_core.data_ptr())
        caller.vcf_stats_device(d_core.data_ptr(), d_out.data_ptr(), 200, n)
        torch.cuda.synchronize()
        cores.append(d_core.cpu().numpy().view(VCF_CORE).copy())
    est = caller.site_stats().copy()
    caller.reset_site_stats()
    for k in range(2):
        d_cts, d_ref = _dev(pile[k * n : (k + 1) * n]), _dev(ref2[k * n : (k + 1) * n + 2])
        d_core = torch.empty(n * 64, dtype=torch.uint8, device=DEV)
        caller.chain_device(d_cts.data_ptr(), d_ref.data_ptr(), 100 + k * n, n, 0, n, d_core.data_ptr(), with_stats=True)
        torch.cuda.synchronize()
        _same_core(d_core.cpu().numpy().view(VCF_CORE), cores[k], "block %d" % k)
    _same_stats(caller.site_stats().copy(), est)


def test_gc_by_coverage(caller):
    """gt_cov_stats.gc_pcent (src/print_vcf.c:394-398): every position that reached the printer counts under [total depth]
    [G+C of its 100-base bin]; bins with an N (255) and positions outside the bins are not counted.  Checked against a
    numpy census of the records; the bins themselves against a direct restatement of load_sequence's loop
    (src/read_reference.c:66-104)."""
    from bs_call_amd.caller import gc_bins

    n, x0 = 30_000, 4000
    pile, ref2 = _block(SEED + 77, n, 30, flags=1)
    ref2 = ref2.copy()
    ref2[12_345] = 0  # an N inside one bin
    codes = np.zeros(x0 - 1 + n - 150, dtype=np.uint8)  # the contig ends 150 positions before the block does: no bin there
    codes[x0 - 1 :] = ref2[: n - 150]
    codes[x0 - 1 : x0 + 6] = 0  # the contig's first A/C/G/T base is 7 positions into the block
    start, bins = gc_bins(codes)
    # restatement: first valid base, then bins of 100
    k = int(np.argmax((codes >= 1) & (codes <= 4)))
    assert start == k + 1 == x0 + 7
    exp_bins = []
    for b0 in range(k, len(codes) - 99, 100):
        w = codes[b0 : b0 + 100]
        exp_bins.append(int(((w == 2) | (w == 3)).sum()) if ((w >= 1) & (w <= 4)).all() else 255)
    assert bins.tolist() == exp_bins and 255 in exp_bins
    d_bins = _dev(bins)
    caller.reset_site_stats()
    caller.set_gc_bins(d_bins.data_ptr(), len(bins), start)
    try:
        got, gst, _ = _fused_keep(caller, pile, ref2, x0, [(0, 11_111), (11_111, 18_889)])
        tab = caller.gc_stats()
    finally:
        caller.set_gc_bins(None, 0, 0)
    depth = pile["counts"].reshape(n, 16).sum(axis=1)
    exp = np.zeros((4096, 101), dtype=np.uint64)
    for i in np.nonzero(got["pos"] != 0)[0]:
        pos = x0 + int(i)
        if pos < start:
            continue
        bn = (pos - start) // 100
        if bn < len(bins) and bins[bn] <= 100:
            exp[min(int(depth[i]), 4095), bins[bn]] += 1
    assert (tab == exp).all() and int(tab.sum()) > 20_000
    assert int(tab.sum(axis=1)[: 4096].sum()) <= int(gst["cov"][:, 0].sum())  # a subset of the "All" coverage column
    # switched off: nothing more is added
    _fused_keep(caller, pile, ref2, x0, [(0, n)])
    assert (caller.gc_stats() == tab).all()


def _fused_keep(c, pile, ref2, x, windows):
    """_fused without the statistics reset (the caller set up what it wants accumulated)."""
    nb = len(pile)
    core = np.zeros(nb, dtype=VCF_CORE)
    for first, n in windows:
        lc, rc, lr = min(2, first), min(2, nb - first - n), min(4, first)
        d_cts = _dev(pile[first - lc : first + n + rc])
        d_ref = _dev(ref2[first - lr : first + n + 2])
        d_core = torch.full((n * 64,), 0xA5, dtype=torch.uint8, device=DEV)
        c.chain_device(d_cts.data_ptr(), d_ref.data_ptr(), x, nb, first, n, d_core.data_ptr(), False, 1, 0xFFFFFFFF, None, True, None)
        torch.cuda.synchronize()
        core[first : first + n] = d_core.cpu().numpy().view(VCF_CORE)
    return core, c.site_stats().copy(), c.stats()


def _adversarial(rng, n):
    """Uniformly random class counts / qualities, depths from 1 to several thousand, N runs, dbSNP flags, uncovered stretches."""
    pile = np.zeros(n, dtype=B.PILEUP)
    depth_scale = rng.choice([1, 3, 10, 40, 400], size=n)
    mask = rng.random((n, 2, 8)) < rng.choice([0.1, 0.3, 0.6, 1.0], size=(n, 1, 1))
    cnt = (rng.integers(0, 8, size=(n, 2, 8)) * depth_scale[:, None, None] * mask).astype(np.uint32)
    cnt[rng.random(n) < 0.05] = 0  # uncovered positions
    pile["counts"] = cnt
    tot = cnt.sum(axis=1)
    pile["n"] = tot.sum(axis=1)
    meanq = rng.integers(20, 44, size=(n, 8))
    pile["quality"] = np.minimum((tot * meanq).astype(np.float32), 43.0 * tot)
    pile["mapq2"] = (pile["n"] * rng.choice([0, 1, 400, 1521, 3600], size=n)).astype(np.float32)
    ref2 = rng.integers(1, 5, size=n + 2).astype(np.uint8)
    for s in rng.integers(0, n - 50, 40):
        ref2[s : s + int(rng.integers(1, 40))] = 0
    flags = rng.choice([0, 0, 0, 1, 3], size=n).astype(np.uint8)
    return pile, ref2, flags


def _oracle_chain(oracle, tables, libm_exact, pile, ref2, x, dbsnp=None, all_positions=False, reg=(1, 0xFFFFFFFF)):
    """The reference's calc threads followed by its print thread, restated: (VCF_CORE[n], SITE_STATS record, counters)."""
    n = len(pile)
    gtm, skip = oracle.call_sites(pile, ref2[:n], tables, oracle.LIBM if libm_exact else oracle.BSM, -8)
    stats = np.zeros(1, dtype=SITE_STATS)
    core = oracle.vcf_block_stats(gtm, skip, ref2, x, stats, np.zeros(2, dtype=np.uint32), tables.lfact_store, all_positions, reg[0],
                                  reg[1], dbsnp)
    cov = skip == 0
    cnt = {"sites": n, "covered": int(cov.sum()), "gt_hist": np.bincount(gtm["max_gt"][cov], minlength=10).tolist(),
           "het_calls": int(np.array(B.GT_HET)[gtm["max_gt"][cov]].sum())}
    return core, stats[0], cnt


def test_fused_vs_oracle_on_adversarial_pileups(caller, oracle, tables, libm_exact):
    """The adversarial pile-ups of the test below — ties, depths beyond the 4 096-row coverage table, N runs, dbSNP flags,
    mostly heterozygous calls — through the FUSED chain in windows of every alignment, against the CPU oracle directly
    (orc_call_sites + orc_vcf_block_stats): records, statistics, counters."""
    if not libm_exact:
        pytest.skip("host libm differs from the replica: the record bytes go through exp/log")
    pile, ref2, flags = _adversarial(np.random.default_rng(20261004), 150_000)
    n = len(pile)
    wins, first = [], 0
    for w in [1, 59, 60, 61, 3_000, 17_001, 40_020, 2, 33_333]:
        wins.append((first, w))
        first += w
    wins.append((first, n - first))
    for kw in (dict(dbsnp=flags), dict(all_positions=True), dict(reg=(777 + 20_000, 777 + 90_000), dbsnp=flags)):
        ecore, est, ecnt = _oracle_chain(oracle, tables, libm_exact, pile, ref2, 777, **kw)
        got, gst, gcnt = _fused(caller, pile, ref2, 777, wins, **kw)
        _same_core(got, ecore, "fused vs oracle, adversarial %s" % (list(kw),))
        _same_stats(gst, est)
        assert gcnt == ecnt


@pytest.mark.parametrize("cov,n", [(200, 60_000), (1400, 6_000)])
def test_fused_vs_oracle_deep_coverage(caller, oracle, tables, libm_exact, cov, n):
    """200x (BASELINE.json configs[3]) and 1 400x blocks — the exp range beyond -512, lgamma in Fisher's test, CpG cytosines
    beyond the LDS and HBM pair tables — through the fused chain against the oracle directly."""
    if not libm_exact:
        pytest.skip("host libm differs from the replica: the record bytes go through exp/log")
    pile, ref2 = _block(SEED + 40 + cov, n, cov)
    ecore, est, ecnt = _oracle_chain(oracle, tables, libm_exact, pile, ref2, 50)
    got, gst, gcnt = _fused(caller, pile, ref2, 50, [(0, n // 3), (n // 3, n - n // 3)])
    _same_core(got, ecore, "fused vs oracle %dx" % cov)
    _same_stats(gst, est)
    assert gcnt == ecnt


def fuzz_round(c, oracle, tables, flav, seed, rnd, always_oracle=False):
    """One round of tools/fuzz_chain.py: a random block (WGBS-like from the generator, or adversarial class counts), random
    window cuts, random printing parameters — fused against the three unfused kernels and (every fourth round, or always)
    against the CPU oracle.  Raises AssertionError on a difference."""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 59, 60, 61, 119, 500, 5_000, 60_000, 200_000]))
    if always_oracle and n > 60_000:
        n = 60_000
    x0 = int(rng.choice([1, 2, 3, 1000, 4_000_000_000 - 300_000]))
    kind = rng.random()
    if kind < 0.5:
        pile, ref2 = B.synth_pileup_host(seed, x0 + 7, n + 2, int(rng.choice([1, 5, 30, 200, 600])), int(rng.integers(0, 2)))
        pile = pile[:n]
    else:
        pile = np.zeros(n, dtype=B.PILEUP)
        scale = rng.choice([1, 1, 3, 10, 40, 400], size=n)
        mask = rng.random((n, 2, 8)) < rng.choice([0.1, 0.3, 0.6, 1.0], size=(n, 1, 1))
        cnt = (rng.integers(0, 8, size=(n, 2, 8)) * scale[:, None, None] * mask).astype(np.uint32)
        cnt[rng.random(n) < rng.choice([0.0, 0.05, 0.5])] = 0
        pile["counts"] = cnt
        tot = cnt.sum(axis=1)
        pile["n"] = tot.sum(axis=1)
        pile["quality"] = np.minimum((tot * rng.integers(20, 44, size=(n, 8))).astype(np.float32), 43.0 * tot)
        pile["mapq2"] = (pile["n"] * rng.choice([0, 1, 400, 1521, 3600], size=n)).astype(np.float32)
        ref2 = rng.integers(0 if rng.random() < 0.3 else 1, 5, size=n + 2).astype(np.uint8)
    flags = rng.choice([0, 0, 0, 1, 3], size=n).astype(np.uint8) if rng.random() < 0.5 else None
    allp = bool(rng.integers(0, 2)) if rng.random() < 0.3 else False
    reg = (1, 0xFFFFFFFF) if rng.random() < 0.7 else (x0 + n // 4, x0 + 3 * n // 4)
    cuts = sorted(set(int(v) for v in rng.integers(0, n + 1, int(rng.integers(0, 6)))) | {0, n})
    wins = [(a, b - a) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    what = "seed=%d n=%d x0=%d kind=%.2f wins=%s" % (seed, n, x0, kind, wins)
    got, gst, gcnt = _fused(c, pile, ref2, x0, wins, dbsnp=flags, all_positions=allp, reg=reg)
    if not always_oracle:
        exp, est, ecnt = _unfused(c, pile, ref2, x0, dbsnp=flags, all_positions=allp, reg=reg)
        _same_core(got, exp, "fused vs unfused " + what)
        _same_stats(gst, est)
        assert gcnt == ecnt, what
    if always_oracle or (rnd % 4 == 0 and n <= 60_000):
        gtm, skip = oracle.call_sites(pile, ref2[:n], tables, flav, -4)
        st = np.zeros(1, dtype=SITE_STATS)
        ocore = oracle.vcf_block_stats(gtm, skip, ref2, x0, st, np.zeros(2, dtype=np.uint32), tables.lfact_store, allp, reg[0], reg[1], flags)
        _same_core(got, ocore, "fused vs oracle " + what)
        _same_stats(gst, st[0])
    return n, len(wins)


def test_fuzz_slice_vs_oracle(caller, oracle, tables, libm_exact):
    """A seeded, time-bounded slice of tools/fuzz_chain.py with EVERY round compared with the CPU oracle directly."""
    import time

    if not libm_exact:
        pytest.skip("host libm differs from the replica: the record bytes go through exp/log")
    t_end, rnd = time.time() + 45.0, 0
    while time.time() < t_end or rnd < 20:
        fuzz_round(caller, oracle, tables, oracle.LIBM, 3_000_017 + rnd, rnd, always_oracle=True)
        rnd += 1
    assert rnd >= 20


def test_fused_equals_unfused_on_adversarial_pileups(caller):
    """Uniformly random class counts / qualities (not WGBS-like: every class combination, depths from 1 to several thousand
    — beyond the 4 096-row coverage table —, near-exact likelihood ties, most calls heterozygous), random reference codes
    with N runs, random dbSNP flags, uncovered stretches: records and statistics of the fused chain, walked in windows of
    every alignment, equal the three unfused kernels'."""
    pile, ref2, flags = _adversarial(np.random.default_rng(20261004), 150_000)
    n = len(pile)
    assert int(pile["n"].max()) > 4096
    exp, est, ecnt = _unfused(caller, pile, ref2, 777, dbsnp=flags)
    wins, first = [], 0
    for w in [1, 59, 60, 61, 3_000, 17_001, 40_020, 2, 33_333]:
        wins.append((first, w))
        first += w
    wins.append((first, n - first))
    got, gst, gcnt = _fused(caller, pile, ref2, 777, wins, dbsnp=flags)
    _same_core(got, exp, "adversarial pile-ups")
    _same_stats(gst, est)
    assert gcnt == ecnt
    het = np.array(B.GT_HET)[got["gt"]] & (got["pos"] != 0)
    assert het.sum() > 0.2 * n and int(gst["cov"][4095, 0]) > 0  # mostly heterozygous calls; the coverage table's last row is in use


def test_fused_many_heterozygous_calls_per_wave(caller, oracle, tables, libm_exact):
    """4 M adversarial positions, 85 % of the calls heterozygous: every resident wave lists ~700 of them, more than its list
    holds (csrc/fused.hip F_HET_CAP 512) — the tile loop's epochs (Fisher pass, list reused, tiles resumed) against the oracle."""
    if not libm_exact:
        pytest.skip("host libm differs from the replica: the record bytes go through exp/log")
    pile, ref2, flags = _adversarial(np.random.default_rng(77), 4_000_000)
    n = len(pile)
    ecore, est, ecnt = _oracle_chain(oracle, tables, libm_exact, pile, ref2, 777, dbsnp=flags)
    assert ecnt["het_calls"] > 0.7 * n
    got, gst, gcnt = _fused(caller, pile, ref2, 777, [(0, n)], dbsnp=flags)
    _same_core(got, ecore, "fused vs oracle, 4 M mostly heterozygous")
    _same_stats(gst, est)
    assert gcnt == ecnt
