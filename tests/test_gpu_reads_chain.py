"""bsc_reads_chain_device (reads in, records out: HOT LOOP A fused into the chain kernel, csrc/fused.hip READS = true)
against the CPU oracle — orc_accumulate -> orc_call_sites -> orc_vcf_block_stats = the reference's process, calc and print
threads (src/call_genotypes.c:180-226 -> :43-115 -> src/print_vcf.c:32-594) — every byte of every bsc_vcf_core record and of
the packed half records (MC8 / AMQ / MQ), every counter, every integer of the statistics."""
import numpy as np
import pytest
import torch

import bs_call_amd as B
from bs_call_amd.abi import SITE_STATS, VCF_CORE

from test_gpu_chain import _dev, _same_core, _same_stats

pytestmark = pytest.mark.gpu
SEED = 88172645463325252
DEV = "cuda:0"
AUX = np.dtype([("counts", "<u4", (8,)), ("qual", "u1", (8,)), ("mq", "<i4"), ("aq", "<i4"), ("max_gt", "u1"), ("rs_found", "u1"),
                ("_pad", "u1", (14,))])


@pytest.fixture(scope="module", params=["two_kernels", "one_kernel"])
def caller(request):
    """Every test of this module in both forms of the reads -> records path (bsc_set_reads_fused): the pile-up through HBM
    (accumulate kernel, then the pile-up-in chain: the default) and the reads-in chain kernel."""
    c = B.SiteCaller()
    c.set_reads_fused(request.param == "one_kernel")
    yield c
    c.close()


def _block(seed, x0, n, cov):
    tpl, seq = B.synth_reads_host(seed, x0, n, cov)
    x = max(1, x0 - 2)
    y = int(max((tpl["pos"] + tpl["len"]).max(), x0)) - 1 if len(tpl) else x0
    return tpl, seq, x, y


def _reads_chain(c, tpl, seq, x, y, ref2, dbsnp=None, all_positions=False, reg=(1, 0xFFFFFFFF), with_stats=True, aux=True):
    n = y - x + 1
    d_tpl = _dev(tpl) if len(tpl) else torch.zeros(64, dtype=torch.uint8, device=DEV)
    d_seq = _dev(seq) if len(seq) else torch.zeros(64, dtype=torch.uint8, device=DEV)
    d_ref = _dev(ref2)
    d_db = None if dbsnp is None else _dev(dbsnp)
    d_core = torch.full((n * 64,), 0xA5, dtype=torch.uint8, device=DEV)
    d_aux = torch.full((n * 64,), 0x5A, dtype=torch.uint8, device=DEV) if aux else None
    c.reset_stats()
    c.reset_site_stats()
    c.reads_chain_device(d_tpl.data_ptr(), len(tpl), d_seq.data_ptr(), len(seq), x, y, d_ref.data_ptr(), d_core.data_ptr(),
                         None if d_aux is None else d_aux.data_ptr(), all_positions, reg[0], reg[1],
                         None if d_db is None else d_db.data_ptr(), with_stats, None)
    c.block_status(None)
    torch.cuda.synchronize()
    core = d_core.cpu().numpy().view(VCF_CORE).copy()
    a = None if d_aux is None else d_aux.cpu().numpy().view(AUX).copy()
    return core, a, c.site_stats().copy(), c.stats()


def _oracle_chain(oracle, tables, libm_exact, tpl, seq, x, y, ref2, dbsnp=None, all_positions=False, reg=(1, 0xFFFFFFFF)):
    n = y - x + 1
    rc, pile = oracle.accumulate(tpl, seq, x, y, 20)
    assert rc == 0
    gtm, skip = oracle.call_sites(pile, ref2[:n], tables, oracle.LIBM if libm_exact else oracle.BSM, -8)
    stats = np.zeros(1, dtype=SITE_STATS)
    carry = np.zeros(2, dtype=np.uint32)
    core = oracle.vcf_block_stats(gtm, skip, ref2, x, stats, carry, tables.lfact_store, all_positions, reg[0], reg[1], dbsnp)
    aux = np.zeros(n, dtype=AUX)
    has = core["pos"] != 0
    aux["counts"][has] = gtm["counts"][has]
    aux["qual"][has] = gtm["qual"][has]
    aux["mq"][has], aux["aq"][has], aux["max_gt"][has] = gtm["mq"][has], gtm["aq"][has], gtm["max_gt"][has]
    if dbsnp is not None:
        aux["rs_found"][has] = dbsnp[has]
    return core, aux, stats[0], gtm, skip


@pytest.mark.parametrize("cov,n,x0", [(30, 120_000, 9_000), (10, 30_000, 5), (200, 6_000, 777_777), (30, 64, 100), (30, 1, 50),
                                      (30, 59, 3), (30, 121, 1), (300, 3_000, 40), (3000, 700, 40)])
def test_reads_chain_vs_oracle(caller, oracle, tables, libm_exact, cov, n, x0):
    if not libm_exact:
        pytest.skip("host libm differs from the replica: the record bytes go through exp/log")
    rng = np.random.default_rng(cov * 1000 + n)
    tpl, seq, x, y = _block(SEED + 700 + cov + n, x0, n, cov)
    sz = y - x + 1
    ref2 = B.synth_ref_host(SEED + 700 + cov + n, x, sz + 2)
    db = rng.choice([0, 1, 3], size=sz, p=[0.9, 0.05, 0.05]).astype(np.uint8)
    for kw in (dict(), dict(all_positions=True), dict(reg=(x + sz // 4, x + sz // 2)), dict(dbsnp=db)):
        ecore, eaux, est, gtm, skip = _oracle_chain(oracle, tables, libm_exact, tpl, seq, x, y, ref2, **kw)
        core, aux, gst, cnt = _reads_chain(caller, tpl, seq, x, y, ref2, **kw)
        _same_core(core, ecore, "reads chain %s" % (kw.keys(),))
        assert aux.tobytes() == eaux.tobytes(), [f for f in AUX.names if aux[f].tobytes() != eaux[f].tobytes()]
        _same_stats(gst, est)
        assert cnt["sites"] == sz and cnt["covered"] == int((skip == 0).sum())
        assert cnt["gt_hist"] == np.bincount(gtm["max_gt"][skip == 0], minlength=10).tolist()


def test_reads_chain_equals_accumulate_then_chain(caller):
    """The same block through the unfused route on the device (bsc_accumulate_device -> bsc_chain_device): same bytes, and
    the counts of a block whose forward counts exceed a byte (the scratch-line path of the heterozygous calls); at 3 000x a
    tile sees more reads than a packed pile-up cell can count (accdev.h ACC_PACK_MAX): both kernels take the unpacked walk."""
    for cov, n in ((30, 400_000), (1400, 4_000), (3000, 1_500)):
        tpl, seq, x, y = _block(SEED + 800 + cov, 5_000, n, cov)
        sz = y - x + 1
        ref2 = B.synth_ref_host(SEED + 800 + cov, x, sz + 2)
        core, aux, gst, cnt = _reads_chain(caller, tpl, seq, x, y, ref2)
        d_tpl, d_seq, d_ref = _dev(tpl), _dev(seq), _dev(ref2)
        pad = (sz + 63) // 64 * 64
        d_cts = torch.zeros(pad * 104 + 256, dtype=torch.uint8, device=DEV)
        d_core = torch.empty(sz * 64, dtype=torch.uint8, device=DEV)
        caller.reset_stats()
        caller.reset_site_stats()
        caller.accumulate_device(d_tpl.data_ptr(), len(tpl), d_seq.data_ptr(), len(seq), x, y, d_cts.data_ptr(), None)
        caller.block_status(None)
        caller.chain_device(d_cts.data_ptr(), d_ref.data_ptr(), x, sz, 0, sz, d_core.data_ptr(), with_stats=True)
        torch.cuda.synchronize()
        _same_core(core, d_core.cpu().numpy().view(VCF_CORE), "unfused route %dx" % cov)
        _same_stats(gst, caller.site_stats())
        assert cnt == caller.stats()
        if cov == 1400:
            assert int(aux["counts"].max()) > 600  # deep enough for forward counts beyond a byte


def test_reads_chain_bad_template_and_empty_block(caller):
    tpl, seq, x, y = _block(SEED + 900, 20_000, 5_000, 30)
    sz = y - x + 1
    ref2 = B.synth_ref_host(SEED + 900, x, sz + 2)
    bad = tpl.copy()
    bad["orientation"][7] = 3
    with pytest.raises(B.BscError) as e:
        _reads_chain(caller, bad, seq, x, y, ref2)
    assert "template 7" in str(e.value)
    core, aux, st, cnt = _reads_chain(caller, tpl[:0], seq[:0], x, x + 199, ref2[:202])
    assert not core.view(np.uint8).any() and not aux.view(np.uint8).any() and cnt["covered"] == 0


def test_reads_chain_many_heterozygous_calls_per_wave(caller, oracle, tables, libm_exact):
    """Reads whose bases are random (qualities kept): most positions are called heterozygous, ~700 per resident wave on 4 M
    positions — more than a wave's list of 80-byte entries holds (csrc/fused.hip F_HET_CAP 512), so the tile loop runs in
    several epochs: Fisher pass, list reused, accumulate resumed.  Against the oracle, every byte."""
    if not libm_exact:
        pytest.skip("host libm differs from the replica: the record bytes go through exp/log")
    tpl, seq, x, y = _block(SEED + 1000, 3_000, 4_000_000, 30)
    rng = np.random.default_rng(31)
    seq = seq ^ rng.integers(0, 4, size=len(seq), dtype=np.uint8)
    sz = y - x + 1
    ref2 = B.synth_ref_host(SEED + 1000, x, sz + 2)
    ecore, eaux, est, gtm, skip = _oracle_chain(oracle, tables, libm_exact, tpl, seq, x, y, ref2)
    het = np.array(B.GT_HET)[gtm["max_gt"][skip == 0]].sum()
    assert het > 0.6 * sz, het
    core, aux, gst, cnt = _reads_chain(caller, tpl, seq, x, y, ref2)
    _same_core(core, ecore, "reads chain, mostly heterozygous")
    assert aux.tobytes() == eaux.tobytes(), [f for f in AUX.names if aux[f].tobytes() != eaux[f].tobytes()]
    _same_stats(gst, est)
    assert cnt["het_calls"] == int(het)


def test_summary_workspace_that_cannot_be_allocated_falls_back_quietly(oracle, tables, libm_exact, monkeypatch):
    """The two-kernel form wants 88 bytes per position of HBM; when that allocation fails the call must run the one-kernel form —
    with the same records, no error, and no stale "out of memory" surfacing at the next launch check (a failed hipMalloc stays
    behind as the runtime's last error).  bsc_debug_fail_summary_alloc makes the allocation fail for real."""
    tpl, seq, x, y = _block(SEED + 91, 4_000, 30_000, 30)
    ref2 = B.synth_ref_host(SEED + 91, x, y - x + 3)
    with B.SiteCaller() as c:
        want = _reads_chain(c, tpl, seq, x, y, ref2)
        c.debug_fail_summary_alloc(True)
        got = _reads_chain(c, tpl, seq, x, y, ref2)
        c.debug_fail_summary_alloc(False)
        again = _reads_chain(c, tpl, seq, x, y, ref2)
    for a in (got, again):
        assert a[0].tobytes() == want[0].tobytes() and a[1].tobytes() == want[1].tobytes()
        _same_stats(a[2], want[2])  # (the methylation profiles are float sums: equal up to the order of the additions)
    exp = _oracle_chain(oracle, tables, libm_exact, tpl, seq, x, y, ref2)
    assert got[0].tobytes() == exp[0].tobytes()


def test_class_mean_qualities_at_their_rounding_boundaries(caller, oracle, tables, libm_exact):
    """Rounded mean qualities of a class (src/call_genotypes.c:49-53) where the mean sits on, or as close as its depth allows to,
    k + 1/2: n single-base reads of qualities k and k + 1, n / 2 (or (n +- 1) / 2) of them the higher one — through the accumulate
    kernel's summary form (or the one-kernel form) against the oracle's f32 quotient -> f64 + 0.5 -> f32 -> floorf, and against the
    mean rounded half up in exact arithmetic; every AMQ byte, at depths from 1 to 600."""
    if not libm_exact:
        pytest.skip("host libm differs from the replica: the record bytes go through exp/log")
    rng = np.random.default_rng(606)
    n_pos, x = 4000, 5000
    depth = rng.integers(1, 601, size=n_pos)
    depth[:64] = np.arange(1, 65)
    k = rng.integers(20, 43, size=n_pos)
    m = np.clip(depth // 2 + rng.integers(-1, 2, size=n_pos), 0, depth)  # reads with quality k + 1
    strand = rng.integers(0, 3, size=n_pos)
    base = rng.integers(0, 4, size=n_pos)
    nt = int(depth.sum())
    tpl = np.zeros(nt, dtype=B.TEMPLATE)
    pos = np.repeat(np.arange(n_pos), depth)
    first = np.concatenate(([0], np.cumsum(depth)[:-1]))
    rank = np.arange(nt) - np.repeat(first, depth)
    q = np.repeat(k, depth) + (rank < np.repeat(m, depth))
    seq = (np.repeat(base, depth) | (q << 2)).astype(np.uint8)
    tpl["pos"][:, 0] = x + 2 + pos
    tpl["len"][:, 0] = 1
    tpl["off"][:, 0] = np.arange(nt)
    tpl["mapq"][:, 0] = 60
    tpl["bs_strand"] = np.repeat(strand, depth)
    tpl["orientation"] = rank & 1
    y = x + 2 + n_pos - 1
    ref2 = rng.integers(1, 5, size=y - x + 3).astype(np.uint8)
    ecore, eaux, est, gtm, skip = _oracle_chain(oracle, tables, libm_exact, tpl, seq, x, y, ref2, all_positions=True)
    core, aux, gst, cnt = _reads_chain(caller, tpl, seq, x, y, ref2, all_positions=True)
    want = np.floor((2 * (k * depth + m) + depth) / (2 * depth)).astype(int)  # the mean, rounded half up, in exact arithmetic
    cls = np.array([[0, 1, 2, 3], [0, 5, 2, 7], [4, 1, 6, 3]])[strand, base]
    assert (gtm["qual"][2 + np.arange(n_pos), cls] == want).all()
    assert (aux["qual"] == eaux["qual"]).all() and aux.tobytes() == eaux.tobytes()
    _same_core(core, ecore, "records")


def test_position_deeper_than_the_summaries_carry(caller, oracle, tables, libm_exact):
    """70 000 reads of one class on one position: more than the 16-bit counts of a site summary hold.  The accumulate kernel flags
    the block (counters[BSC_CNT_DEEP]); of the two chain launches queued behind it the summary-in one stands back and the reads-in
    twin does the work — same records as the oracle, statistics added once, and the next (ordinary) block goes the usual way."""
    if not libm_exact:
        pytest.skip("host libm differs from the replica: the record bytes go through exp/log")
    x, n_pos = 9000, 700
    tpl0, seq0 = B.synth_reads_host(SEED + 4, x + 2, n_pos - 110, 20)
    deep = np.zeros(70_000, dtype=B.TEMPLATE)
    deep["pos"][:, 0] = x + 300
    deep["len"][:, 0] = 1
    deep["off"][:, 0] = len(seq0) + np.arange(len(deep))
    deep["mapq"][:, 0] = 10
    deep["bs_strand"] = 1
    deep["orientation"] = np.arange(len(deep)) & 1
    tpl = np.concatenate([tpl0, deep])
    seq = np.concatenate([seq0, np.full(len(deep), 3 | (30 << 2), dtype=np.uint8)])  # T on C2T reads: class 7
    y = x + n_pos - 1
    ref2 = B.synth_ref_host(SEED + 4, x, n_pos + 2)
    ecore, eaux, est, gtm, skip = _oracle_chain(oracle, tables, libm_exact, tpl, seq, x, y, ref2)
    assert int(gtm["counts"][300, 7]) >= 70_000
    core, aux, gst, cnt = _reads_chain(caller, tpl, seq, x, y, ref2)
    _same_core(core, ecore, "deep block")
    assert aux.tobytes() == eaux.tobytes()
    _same_stats(gst, est)
    # an ordinary block afterwards: the flag is per block
    ecore, eaux, est, gtm, skip = _oracle_chain(oracle, tables, libm_exact, tpl0, seq0, x, y, ref2)
    core, aux, gst, cnt = _reads_chain(caller, tpl0, seq0, x, y, ref2)
    _same_core(core, ecore, "ordinary block")
    _same_stats(gst, est)
