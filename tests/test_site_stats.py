"""Site statistics (the sum fields of the reference's bs_stats, src/print_vcf.c:382-526): the oracle's sequential
restatement checked against hand-counted cases (CPU), and the device histogram kernel against the oracle (GPU) —
integer fields exactly, the methylation profiles (sums of doubles in a different order) to 1e-12 relative."""
import numpy as np
import pytest

import bs_call_amd as B

SEED = 88172645463325252
INT_FIELDS = [n for n in B.SITE_STATS.names if not n.endswith("_meth")]


def _called_block(oracle, tables, seed, x, n, cov, flags=0):
    pile, ref = B.synth_pileup_host(seed, x, n + 2, cov, flags)
    out, skip = oracle.call_sites(pile[:n], ref[:n], tables, oracle.LIBM, -8)
    return out, skip, ref


def _oracle_stats(oracle, tables, blocks, **kw):
    """blocks: iterable of (gtm, skip, ref, x[, dbsnp]) in position order -> (SITE_STATS record, list of records)."""
    st = np.zeros(1, dtype=B.SITE_STATS)
    carry = np.zeros(2, dtype=np.uint32)
    recs = []
    for blk in blocks:
        gtm, skip, ref, x = blk[:4]
        db = blk[4] if len(blk) > 4 else None
        recs.append(oracle.vcf_block_stats(gtm, skip, ref, x, st, carry, tables.lfact_store, dbsnp=db, **kw))
    return st[0], recs


def test_layout_matches_the_c_struct(oracle):
    assert oracle.lib().orc_sizeof_site_stats() == B.SITE_STATS.itemsize
    assert B.SITE_STATS.fields["mut_counts"][1] == 14 * 8 and B.SITE_STATS.fields["cov"][1] == (14 + 48 + 1024 + 64 + 1536) * 8
    assert B.SITE_STATS_INT_WORDS * 8 == B.SITE_STATS.fields["CpG_ref_meth"][1]


def test_oracle_stats_hand_counted(oracle, tables):
    out, skip, ref = _called_block(oracle, tables, SEED + 5, 2000, 20_000, 30)
    st, (rec,) = _oracle_stats(oracle, tables, [(out, skip, ref, 2000)])
    called = rec["pos"] != 0
    emit = rec["emit"] == 1
    dp_all = out["counts"].sum(axis=1)
    # every position that reaches the printer, by total depth
    assert st["cov"][:, 0].sum() == called.sum()
    assert (np.bincount(np.minimum(dp_all[called], B.abi.COV_CAP - 1), minlength=B.abi.COV_CAP) == st["cov"][:, 0]).all()
    # the reference's alt quirk: every written record is a "SNP", none multi-allelic
    assert st["snps"][0] == emit.sum() and st["snps"][1] == (emit & (rec["flt"] == 0)).sum() and st["multi"].sum() == 0
    assert st["qual"][0].sum() == emit.sum() and (st["qual"][0] == st["qual"][1]).all()
    assert (np.bincount(rec["phred"][emit], minlength=256) == st["qual"][0]).all()
    het = np.isin(rec["gt"], [1, 2, 3, 5, 6, 8])
    for h in (0, 1):
        sel = emit & (het == bool(h))
        assert (np.bincount(rec["flt"][sel] & 31, minlength=32) == st["filter_counts"][h]).all()
        assert (np.bincount(out["mq"][sel], minlength=256) == st["mq_stats"][:, h]).all()
        assert (np.bincount(rec["qd"][sel], minlength=256) == st["qd_stats"][:, h]).all()
    # CpGs: a written '-' strand CG call right after a written '+' strand CG call
    plus = emit & (rec["cg"] == b"C") & np.isin(rec["gt"], [1, 4, 6])
    minus = emit & (rec["cg"] == b"C") & np.isin(rec["gt"], [2, 7, 8])
    pair = minus[1:] & plus[:-1]
    assert st["CpG_ref"][0] + st["CpG_nonref"][0] == pair.sum() > 50
    assert st["qual"][2].sum() + st["qual"][3].sum() == (plus | minus).sum()
    # mutation counts: calls that differ from the reference base
    assert st["mut_counts"][:, 0].sum() == (emit & (rec["alt"] != b"") & (rec["ref_code"] != 0)
                                            & ~((rec["n_gl"] == 6))).sum()
    # each posterior sums to one: the profiles add up to the number of CpG cytosines with informative reads
    a = np.where(plus, out["counts"][:, 5] + out["counts"][:, 7], out["counts"][:, 6] + out["counts"][:, 4])
    n_inf = ((plus | minus) & (a > 0)).sum()
    assert abs(st["CpG_ref_meth"][0].sum() + st["CpG_nonref_meth"][0].sum() - n_inf) < 1e-6 * n_inf
    assert st["dbSNP_sites"].sum() == 0 and st["indels"].sum() == 0


def test_oracle_carry_across_blocks(oracle, tables):
    """Two half blocks give the same sums as the whole block when the printer's CpG state is carried (the flush at a
    block end repeats the last genotype for two positions, so only blocks cut where that cannot matter compare)."""
    out, skip, ref = _called_block(oracle, tables, SEED + 6, 500, 6_000, 30)
    whole, _ = _oracle_stats(oracle, tables, [(out, skip, ref, 500)], all_positions=True)
    assert whole["CpG_ref"][0] + whole["CpG_nonref"][0] > 10
    cut = 3_000
    halves, _ = _oracle_stats(oracle, tables, [(out[:cut], skip[:cut], ref[:cut + 2], 500),
                                               (out[cut:], skip[cut:], ref[cut:], 500 + cut)], all_positions=True)
    # context-dependent fields may differ at the cut (CG status of the two positions before it); depth tables may not
    assert (whole["cov"][:, 0] == halves["cov"][:, 0]).all()
    assert abs(int(whole["snps"][0]) - int(halves["snps"][0])) == 0


def _compare(got, exp):
    for f in INT_FIELDS:
        assert (got[f] == exp[f]).all(), (f, np.argwhere(got[f] != exp[f])[:5])
    for f in ("CpG_ref_meth", "CpG_nonref_meth"):
        assert np.allclose(got[f], exp[f], rtol=1e-12, atol=1e-12), f


@pytest.mark.gpu
def test_device_stats_parity(oracle, tables, libm_exact):
    rng = np.random.default_rng(11)
    with B.SiteCaller() as c:
        for cov, n, x in ((30, 150_000, 4_000), (10, 40_000, 77), (300, 6_000, 1_000_000), (1200, 700, 50)):
            out, skip, ref = _called_block(oracle, tables, SEED + 100 + cov, x, n, cov)
            ref = ref.copy()
            ref[rng.integers(0, n + 2, size=n // 60)] = 0
            skip = skip.copy()
            skip[rng.integers(0, n, size=n // 25)] = 1
            db = rng.choice([0, 1, 3], size=n, p=[0.9, 0.05, 0.05]).astype(np.uint8)
            for kw in (dict(), dict(all_positions=True), dict(reg_start=x + 100, reg_stop=x + n // 2), dict(dbsnp=db)):
                st = np.zeros(1, dtype=B.SITE_STATS)
                carry = np.zeros(2, dtype=np.uint32)
                okw = {k: v for k, v in kw.items()}
                exp_rec = oracle.vcf_block_stats(out, skip, ref, x, st, carry, tables.lfact_store, **okw)
                got_rec = c.vcf_records(out, skip, ref, x, **kw)
                if libm_exact:
                    assert got_rec.tobytes() == exp_rec.tobytes()
                c.reset_site_stats()
                c.vcf_stats(exp_rec, out, dbsnp=kw.get("dbsnp"))  # same records in: isolates the statistics kernel
                _compare(c.site_stats(), st[0])
            # strided gt_vcf records behind the statistics
            raw = np.zeros((n, 208), dtype=np.uint8)
            raw[:, :200] = out.view(np.uint8).reshape(n, 200)
            st = np.zeros(1, dtype=B.SITE_STATS)
            exp_rec = oracle.vcf_block_stats(out, skip, ref, x, st, np.zeros(2, dtype=np.uint32), tables.lfact_store)
            c.reset_site_stats()
            c.vcf_stats(exp_rec, raw)
            _compare(c.site_stats(), st[0])


@pytest.mark.gpu
def test_device_stats_blocks_in_order_carry_the_pending_cytosine(oracle, tables):
    """A contig passed as consecutive blocks: the sums accumulate in the context, and a CpG whose C ends one block and
    whose G starts the next is counted (the reference's prev_cpg_x is static)."""
    n, x = 40_000, 3_000
    out, skip, ref = _called_block(oracle, tables, SEED + 300, x, n, 30)
    # find a '+'/'-' CG pair and cut the contig between its two positions
    rec = oracle.vcf_block(out, skip, ref, x, all_positions=True)
    plus = (rec["emit"] == 1) & (rec["cg"] == b"C") & np.isin(rec["gt"], [1, 4, 6])
    minus = (rec["emit"] == 1) & (rec["cg"] == b"C") & np.isin(rec["gt"], [2, 7, 8])
    pairs = np.flatnonzero(plus[:-1] & minus[1:])
    cut = int(pairs[len(pairs) // 2]) + 1
    blocks = [(out[:cut], skip[:cut], ref[:cut + 2], x), (out[cut:], skip[cut:], ref[cut:], x + cut)]
    exp, recs = _oracle_stats(oracle, tables, blocks, all_positions=True)
    with B.SiteCaller() as c:
        for (g, s, r, bx), rc in zip(blocks, recs):
            c.vcf_stats(rc, g)
        got = c.site_stats()
        _compare(got, exp)
        # without the carry the pair would be lost
        c.reset_site_stats()
        c.vcf_stats(recs[0], blocks[0][0])
        c.reset_site_stats()
        c.vcf_stats(recs[1], blocks[1][0])
        lone = c.site_stats()
    second, _ = _oracle_stats(oracle, tables, blocks[1:], all_positions=True)
    _compare(lone, second)
    assert int(exp["CpG_ref"][0] + exp["CpG_nonref"][0]) > 0


@pytest.mark.gpu
def test_device_chain_with_stats(oracle, tables, libm_exact):
    """pile-up -> gt_meth -> VCF records -> statistics, all resident in HBM, on torch's stream."""
    import torch

    n, x, cov = 200_000, 10_000, 30
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    with B.SiteCaller() as c:
        d_cts = torch.empty((n + 2) * 104, dtype=torch.uint8, device=dev)
        d_ref = torch.empty(n + 2, dtype=torch.uint8, device=dev)
        d_out = torch.empty(n * 200, dtype=torch.uint8, device=dev)
        d_skip = torch.empty(n, dtype=torch.uint8, device=dev)
        d_vcf = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        c.synth_device(SEED + 9, x, n + 2, cov, d_cts.data_ptr(), d_ref.data_ptr(), 0, st)
        c.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out.data_ptr(), d_skip.data_ptr(), 200, st)
        c.vcf_records_device(d_out.data_ptr(), 200, d_skip.data_ptr(), d_ref.data_ptr(), n, x, d_vcf.data_ptr(), stream=st)
        c.vcf_stats_device(d_vcf.data_ptr(), d_out.data_ptr(), 200, n, stream=st)
        got = c.site_stats()
    pile, ref = B.synth_pileup_host(SEED + 9, x, n + 2, cov)
    out, skip = oracle.call_sites(pile[:n], ref[:n], tables, oracle.LIBM if libm_exact else oracle.BSM, -8)
    exp, _ = _oracle_stats(oracle, tables, [(out, skip, ref, x)])
    if libm_exact:
        _compare(got, exp)
    else:
        assert got["cov"][:, 0].sum() == exp["cov"][:, 0].sum()
