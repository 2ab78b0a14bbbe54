"""bsc_block_records / bsc_vcf_compact_device: reads in, written records out, packed in position order — against the
oracle chain accumulate -> call -> print_vcf restatement, byte for byte."""
import numpy as np
import pytest

import bs_call_amd as B

pytestmark = pytest.mark.gpu
SEED = 88172645463325252


@pytest.fixture(scope="module")
def caller():
    c = B.SiteCaller()
    yield c
    c.close()


def _block(seed, x0, n, cov):
    tpl, seq = B.synth_reads_host(seed, x0, n, cov)
    x = max(1, x0 - 2)
    y = int(max((tpl["pos"] + tpl["len"]).max(), x0)) - 1 if len(tpl) else x0
    return tpl, seq, x, y


def _expected(oracle, tables, libm_exact, tpl, seq, x, y, ref, **kw):
    rc, pile = oracle.accumulate(tpl, seq, x, y, 20)
    gtm, skip = oracle.call_sites(pile, ref[: y - x + 1], tables, oracle.LIBM if libm_exact else oracle.BSM, -8)
    core = oracle.vcf_block(gtm, skip, ref, x, **kw)
    sel = core["emit"] == 1
    exp = np.zeros(int(sel.sum()), dtype=B.VCF_REC)
    exp["core"] = core[sel]
    exp["counts"] = gtm["counts"][sel]
    exp["qual"] = gtm["qual"][sel]
    exp["mq"], exp["aq"], exp["max_gt"] = gtm["mq"][sel], gtm["aq"][sel], gtm["max_gt"][sel]
    if kw.get("dbsnp") is not None:
        exp["rs_found"] = kw["dbsnp"][sel]
    return exp, gtm, skip, core


def test_block_records_parity(caller, oracle, tables, libm_exact):
    if not libm_exact:
        pytest.skip("host libm differs from the replica: the record bytes go through exp/log")
    rng = np.random.default_rng(21)
    for cov, n, x0 in ((30, 120_000, 9_000), (10, 30_000, 5), (200, 5_000, 777_777), (30, 64, 100), (30, 1, 50)):
        tpl, seq, x, y = _block(SEED + 500 + cov + n, x0, n, cov)
        sz = y - x + 1
        ref = B.synth_ref_host(SEED + 500 + cov + n, x, sz + 2)
        db = rng.choice([0, 1, 3], size=sz, p=[0.9, 0.05, 0.05]).astype(np.uint8)
        for kw in (dict(), dict(all_positions=True), dict(reg_start=x + sz // 4, reg_stop=x + sz // 2), dict(dbsnp=db)):
            exp, gtm, skip, core = _expected(oracle, tables, libm_exact, tpl, seq, x, y, ref, **kw)
            got = caller.block_records(tpl, seq, x, y, ref, **kw)
            assert len(got) == len(exp), (cov, n, kw.keys())
            assert got.tobytes() == exp.tobytes(), (cov, n, kw.keys())
        # positions strictly increasing = position order
        got = caller.block_records(tpl, seq, x, y, ref, all_positions=True)
        assert (np.diff(got["core"]["pos"].astype(np.int64)) > 0).all()


def test_block_records_capacity_stats_and_text(caller, oracle, tables, libm_exact):
    tpl, seq, x, y = _block(SEED + 600, 40_000, 50_000, 30)
    sz = y - x + 1
    ref = B.synth_ref_host(SEED + 600, x, sz + 2)
    exp, gtm, skip, core = _expected(oracle, tables, libm_exact, tpl, seq, x, y, ref)
    small = np.zeros(len(exp) - 1, dtype=B.VCF_REC)
    with pytest.raises(B.BscError) as e:
        caller.block_records(tpl, seq, x, y, ref, out=small)
    assert str(len(exp)) in str(e.value)
    exact = np.zeros(len(exp), dtype=B.VCF_REC)
    caller.reset_site_stats()
    got = caller.block_records(tpl, seq, x, y, ref, out=exact, with_stats=True)
    assert len(got) == len(exp)
    st = caller.site_stats()
    assert int(st["snps"][0]) == len(exp) and int(st["cov"][:, 0].sum()) == int((core["pos"] != 0).sum())
    # an empty block (no reads at all) and a bad template
    assert len(caller.block_records(tpl[:0], seq, x, x + 99, ref[:102])) == 0
    bad = tpl.copy()
    bad["bs_strand"][1] = 5
    with pytest.raises(B.BscError):
        caller.block_records(bad, seq, x, y, ref)
    # the packed record renders the same VCF line as the full gt_meth record
    import ctypes as C

    L = caller._L
    buf1, buf2 = C.create_string_buffer(2048), C.create_string_buffer(2048)
    idx = np.flatnonzero(core["emit"] == 1)
    for j in (0, len(exp) // 2, len(exp) - 1):
        n1 = L.bsc_vcf_format_rec(got[j : j + 1].ctypes.data, b"chrS", None, buf1, 2048)
        c, g = core[idx[j] : idx[j] + 1], gtm[idx[j] : idx[j] + 1]
        n2 = L.bsc_vcf_format(c.ctypes.data, g.ctypes.data, b"chrS", None, buf2, 2048)
        if libm_exact:
            assert n1 == n2 > 0 and buf1.value == buf2.value


def test_compact_device_chain(oracle, tables, libm_exact):
    """pile-up -> gt_meth -> records -> packed records, all in HBM on torch's stream; count read back from the device."""
    import torch

    n, x, cov = 250_000, 10_000, 30
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    with B.SiteCaller() as c:
        d_cts = torch.empty((n + 2) * 104, dtype=torch.uint8, device=dev)
        d_ref = torch.empty(n + 2, dtype=torch.uint8, device=dev)
        d_out = torch.empty(n * 200, dtype=torch.uint8, device=dev)
        d_skip = torch.empty(n, dtype=torch.uint8, device=dev)
        d_vcf = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        d_rec = torch.empty(n * 128, dtype=torch.uint8, device=dev)
        d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        c.synth_device(SEED + 9, x, n + 2, cov, d_cts.data_ptr(), d_ref.data_ptr(), 0, st)
        c.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out.data_ptr(), d_skip.data_ptr(), 200, st)
        c.vcf_records_device(d_out.data_ptr(), 200, d_skip.data_ptr(), d_ref.data_ptr(), n, x, d_vcf.data_ptr(), stream=st)
        c.vcf_compact_device(d_vcf.data_ptr(), d_out.data_ptr(), 200, n, d_rec.data_ptr(), n, d_cnt.data_ptr(), stream=st)
        cnt = int(d_cnt.item())
        got = d_rec[: cnt * 128].cpu().numpy().view(B.VCF_REC)
        # a capacity smaller than the block: counted, not stored
        c.vcf_compact_device(d_vcf.data_ptr(), d_out.data_ptr(), 200, n, d_rec.data_ptr(), 10, d_cnt.data_ptr(), stream=st)
        assert int(d_cnt.item()) == cnt
    pile, ref = B.synth_pileup_host(SEED + 9, x, n + 2, cov)
    gtm, skip = oracle.call_sites(pile[:n], ref[:n], tables, oracle.LIBM if libm_exact else oracle.BSM, -8)
    core = oracle.vcf_block(gtm, skip, ref, x)
    sel = core["emit"] == 1
    if libm_exact:
        assert cnt == int(sel.sum())
        assert got["core"].tobytes() == core[sel].tobytes()
        assert (got["counts"] == gtm["counts"][sel]).all() and (got["qual"] == gtm["qual"][sel]).all()
        assert (got["mq"] == gtm["mq"][sel]).all() and (got["max_gt"] == gtm["max_gt"][sel]).all()


def test_chain_regression_anchor_on_the_device(caller, libm_exact):
    """The committed fixture of the whole chain (tests/golden/chain_regression.json), reproduced by the device path
    alone: reads -> pile-up -> gt_meth -> records -> statistics."""
    import json
    import os

    if not libm_exact:
        pytest.skip("fixture was written on a host whose libm equals the replica")
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chain_regression.json")))
    g = gold["generator"]
    tpl, seq = B.synth_reads_host(g["seed"], g["x0"], g["n_sites"], g["coverage"])
    x, y = gold["block"]["x"], gold["block"]["y"]
    ref = B.synth_ref_host(g["seed"], x, y - x + 3)
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_golden_chain", os.path.join(root, "tools", "make_golden_chain.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    sha = m.digest  # per field: numpy leaves the padding bytes of copied records undefined
    assert sha(tpl) == gold["sha256"]["templates"] and sha(seq) == gold["sha256"]["seq"] and sha(ref) == gold["sha256"]["ref"]
    assert sha(caller.accumulate(tpl, seq, x, y)) == gold["sha256"]["pileup"]
    gtm, skip = caller.call_block(tpl, seq, x, y, ref[: y - x + 1])
    assert sha(gtm) == gold["sha256"]["gt_meth"] and sha(skip) == gold["sha256"]["skip"]
    core = caller.vcf_records(gtm, skip, ref, x)
    assert sha(core) == gold["sha256"]["vcf_core"]
    caller.reset_site_stats()
    caller.vcf_stats(core, gtm)
    st = caller.site_stats()
    assert sha(np.frombuffer(st.tobytes()[: B.SITE_STATS_INT_WORDS * 8], dtype=np.uint64)) == gold["sha256"]["site_stats_int"]
    assert abs(float(st["CpG_ref_meth"][0].sum() + st["CpG_nonref_meth"][0].sum()) - gold["meth_profile_sum"]) < 1e-9
    recs = caller.block_records(tpl, seq, x, y, ref)
    assert sha(recs["core"]) == sha(core[core["emit"] == 1])


def test_block_records_submit_fetch(caller, oracle, tables, libm_exact):
    """The split form (bsc_block_records_submit / _fetch): blocks queued one after the other, inputs recycled right after
    the submit, records = the synchronous call's = the oracle chain's; a block that writes a record for every position
    (-A) after blocks that write half exercises the second copy of the copy-out."""
    blocks = []
    for k, (cov, n, x0, kw) in enumerate(((30, 60_000, 9_000, dict()), (30, 40_000, 200_000, dict(all_positions=True)),
                                          (10, 30_000, 400_000, dict()), (30, 5_000, 500_000, dict(all_positions=True)))):
        tpl, seq, x, y = _block(SEED + 900 + k, x0, n, cov)
        ref = B.synth_ref_host(SEED + 900 + k, x, y - x + 3)
        blocks.append((tpl, seq, x, y, ref, kw))
    sync = [caller.block_records(t, s, x, y, r, **kw).copy() for t, s, x, y, r, kw in blocks]
    if libm_exact:
        for (t, s, x, y, r, kw), got in zip(blocks, sync):
            exp = _expected(oracle, tables, libm_exact, t, s, x, y, r, **kw)[0]
            assert got.tobytes() == exp.tobytes()
    outs = [B.PinnedBuffer(y - x + 1, B.VCF_REC) for _, _, x, y, _, _ in blocks]
    got = []
    for k, (t, s, x, y, r, kw) in enumerate(blocks):
        t2, s2, r2 = t.copy(), s.copy(), r.copy()
        caller.block_records_submit(t2, s2, x, y, r2, outs[k].array, **kw)
        t2[:], s2[:], r2[:] = 0, 0, 0  # the inputs were staged: the caller may recycle them at once
        with pytest.raises(B.BscError):
            caller.block_records_submit(t, s, x, y, r, outs[k].array, **kw)  # one block in flight per context
        got.append(caller.block_records_fetch().copy())
    for a, b in zip(got, sync):
        assert a.tobytes() == b.tobytes()
    with pytest.raises(B.BscError):
        caller.block_records_fetch()
    # a bad template surfaces at the fetch, and the context goes on working
    t, s, x, y, r, kw = blocks[0]
    bad = t.copy()
    bad["bs_strand"][3] = 7
    caller.block_records_submit(bad, s, x, y, r, outs[0].array)
    with pytest.raises(B.BscError) as e:
        caller.block_records_fetch()
    assert "template 3" in str(e.value)
    caller.block_records_submit(t, s, x, y, r, outs[0].array)
    assert caller.block_records_fetch().tobytes() == sync[0].tobytes()
    # the in-place form: no staging copy, the (page-locked) inputs read where they lie; two contexts alternating, block k + 1
    # submitted before block k is fetched — the pipeline of tools/bench_two_contexts.py
    with B.SiteCaller() as c2:
        ctx = (caller, c2)
        pins = []
        for t, s, x, y, r, kw in blocks:
            pt, ps, pr = B.PinnedBuffer(len(t), t.dtype), B.PinnedBuffer(len(s), np.uint8), B.PinnedBuffer(len(r), np.uint8)
            pt.array[:], ps.array[:], pr.array[:] = t, s, r
            pins.append((pt, ps, pr))
        got = []
        for k, (t, s, x, y, r, kw) in enumerate(blocks):
            pt, ps, pr = pins[k]
            ctx[k & 1].block_records_submit(pt.array, ps.array, x, y, pr.array, outs[k].array, inplace=True, **kw)
            if k:
                got.append(ctx[(k - 1) & 1].block_records_fetch().copy())
        got.append(ctx[(len(blocks) - 1) & 1].block_records_fetch().copy())
        for a, b in zip(got, sync):
            assert a.tobytes() == b.tobytes()
