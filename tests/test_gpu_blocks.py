"""bsc_blocks_records: several blocks in ONE launch sequence — the reference's real unit is a block of 10^2 .. 10^7 positions
per call_genotypes_ML (src/get_template_vector.c:141-147), and a GPU block costs a dozen launches whatever its size.  The
batched call must give the bytes of bsc_block_records called on the blocks one after another (records AND site statistics),
which are the bytes of the oracle chain accumulate -> call -> print_vcf restatement."""
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
    x = max(1, x0 - 2)  # process_template_vector, src/process_template.c:24-28
    y = int(max((tpl["pos"] + tpl["len"]).max(), x0)) - 1 if len(tpl) else x0
    return tpl, seq, x, y


def _expected(oracle, tables, libm_exact, tpl, seq, x, y, ref, dbsnp=None):
    rc, pile = oracle.accumulate(tpl, seq, x, y, 20)
    gtm, skip = oracle.call_sites(pile, ref[: y - x + 1], tables, oracle.LIBM if libm_exact else oracle.BSM, -8)
    core = oracle.vcf_block(gtm, skip, ref, x, dbsnp=dbsnp)
    sel = core["emit"] == 1
    exp = np.zeros(int(sel.sum()), dtype=B.VCF_REC)
    exp["core"] = core[sel]
    exp["counts"], exp["qual"] = gtm["counts"][sel], gtm["qual"][sel]
    exp["mq"], exp["aq"], exp["max_gt"] = gtm["mq"][sel], gtm["aq"][sel], gtm["max_gt"][sel]
    if dbsnp is not None:
        exp["rs_found"] = dbsnp[sel]
    return exp


def _split_at_cpg(tpl, seq, x, y, ref):
    """One block cut in two at a reference CpG whose G is the first base of some read — block A ends on the C, block B begins on
    the G (x_B = y_A + 1, B's first position covered): the one place where the printer's carried state (prev_cpg_x,
    src/print_vcf.c:447-455) is looked at across blocks, and where bsc_blocks_records starts a new launch."""
    starts = np.unique(tpl["pos"][:, 0][tpl["pos"][:, 0] > x + 500])
    for p in starts:
        if p < y - 500 and ref[p - 1 - x] == 2 and ref[p - x] == 3:
            left = np.where(tpl["pos"][:, 0] != 0, tpl["pos"][:, 0], tpl["pos"][:, 1])
            a, b = tpl[left < p], tpl[left >= p]
            return [(a, seq, x, int(p) - 1, ref[: p - x + 2]), (b, seq, int(p), y, ref[p - x :])]
    return [(tpl, seq, x, y, ref)]


def _random_blocks(rng, n_blocks, lo, hi, cov_choices=(10, 30), split_every=0):
    """(templates, reads, x, y, reference codes) in genome order with random sizes and gaps; every `split_every`-th block is cut
    in two adjacent ones at a CpG (_split_at_cpg)."""
    blocks, pos = [], 1000
    for i in range(n_blocks):
        n = int(rng.integers(lo, hi + 1))
        cov = int(rng.choice(cov_choices))
        x0 = pos + int(rng.integers(3, 400))
        tpl, seq, x, y = _block(SEED + 9000 + i, x0, n, cov)
        ref = B.synth_ref_host(SEED + 9000 + i, x, y - x + 3)
        blocks += _split_at_cpg(tpl, seq, x, y, ref) if split_every and i % split_every == 1 and n > 2000 else [(tpl, seq, x, y, ref)]
        pos = y
    return blocks


def _one_by_one(caller, blocks, refs, dbs=None, **kw):
    out = []
    for i, (tpl, seq, x, y) in enumerate(blocks):
        out.append(caller.block_records(tpl, seq, x, y, refs[i], dbsnp=None if dbs is None else dbs[i], **kw).copy())
    return out


def _cpg_pairs(st):
    return int(np.asarray(st["CpG_ref"]).reshape(-1)[0]) + int(np.asarray(st["CpG_nonref"]).reshape(-1)[0])


def test_500_random_blocks_batched_equal_one_by_one_equal_oracle(caller, oracle, tables, libm_exact):
    if not libm_exact:
        pytest.skip("host libm differs from the replica: the record bytes go through exp/log")
    rng = np.random.default_rng(404)
    blocks = _random_blocks(rng, 500, 50, 50_000, split_every=5)
    # a mix of sizes: most blocks small (the reference's real distribution is heavy at the low end)
    blocks = [b for i, b in enumerate(blocks) if (b[3] - b[2] < 6_000) or i % 7 == 0 or b[2] == blocks[i - 1][3] + 1
              or (i + 1 < len(blocks) and blocks[i + 1][2] == b[3] + 1)]
    n_adjacent = sum(1 for i in range(1, len(blocks)) if blocks[i][2] == blocks[i - 1][3] + 1)
    assert n_adjacent >= 10
    refs = [b[4] for b in blocks]
    blocks = [b[:4] for b in blocks]
    caller.reset_site_stats()
    single = _one_by_one(caller, blocks, refs, with_stats=True)
    st1 = caller.site_stats().copy()
    caller.reset_site_stats()
    got, per = caller.blocks_records(blocks, refs, with_stats=True)
    st2 = caller.site_stats().copy()
    assert [int(v) for v in per] == [len(s) for s in single]
    assert got.tobytes() == np.concatenate(single).tobytes()
    # the statistics too, the CpG pairs across adjacent blocks included: integers exact; the methylation profiles are sums of
    # doubles in whatever order the device adds them (1e-15 relative)
    for f in B.SITE_STATS.names:
        if f.endswith("_meth"):
            assert np.allclose(st1[f], st2[f], rtol=1e-12, atol=0), f
        else:
            assert st1[f].tobytes() == st2[f].tobytes(), f
    assert _cpg_pairs(st2) > 1000
    # ... and both are the oracle chain's (a sample of the blocks: the oracle walks them on the CPU)
    o = 0
    for i, (tpl, seq, x, y) in enumerate(blocks):
        if i % 9 == 0:
            exp = _expected(oracle, tables, libm_exact, tpl, seq, x, y, refs[i])
            assert got[o : o + int(per[i])].tobytes() == exp.tobytes(), i
        o += int(per[i])


def test_blocks_with_dbsnp_all_positions_and_edge_sizes(caller, oracle, tables, libm_exact):
    if not libm_exact:
        pytest.skip("host libm differs from the replica: the record bytes go through exp/log")
    rng = np.random.default_rng(5)
    # sizes around the tile geometry: 1 position, under a tile, 60 / 62 / 64 boundaries, one large block among small ones
    sizes = [1, 2, 3, 57, 58, 59, 60, 61, 62, 63, 64, 65, 119, 120, 121, 122, 123, 124, 125, 126, 127, 128, 129, 300_000, 5, 40_000, 7]
    blocks, pos = [], 500
    for i, n in enumerate(sizes):
        tpl, seq, x, y = _block(SEED + 31_000 + i, pos + 150, n, 30 if n < 100_000 else 12)
        blocks.append((tpl, seq, x, y))
        pos = y
    blocks.append((blocks[0][0][:0], blocks[0][1], pos + 10, pos + 300))  # a block without reads
    refs = [B.synth_ref_host(SEED + 31_000 + i, x, y - x + 3) for i, (_, _, x, y) in enumerate(blocks)]
    dbs = [rng.choice([0, 1, 3], size=y - x + 1, p=[0.9, 0.05, 0.05]).astype(np.uint8) for _, _, x, y in blocks]
    for kw in (dict(), dict(all_positions=True), dict(reg_start=blocks[5][2], reg_stop=blocks[-3][3] - 1000)):
        single = _one_by_one(caller, blocks, refs, dbs, **kw)
        got, per = caller.blocks_records(blocks, refs, dbsnp=dbs, **kw)
        assert [int(v) for v in per] == [len(s) for s in single], kw
        assert got.tobytes() == np.concatenate(single).tobytes(), kw
    exp = np.concatenate([_expected(oracle, tables, libm_exact, t, s, x, y, refs[i], dbs[i]) for i, (t, s, x, y) in enumerate(blocks)])
    got, per = caller.blocks_records(blocks, refs, dbsnp=dbs)
    assert got.tobytes() == exp.tobytes()


def test_blocks_errors_and_the_one_in_flight_rule(caller):
    tpl, seq, x, y = _block(SEED + 1, 5_000, 3_000, 20)
    tpl2, seq2, x2, y2 = _block(SEED + 2, y + 50, 2_000, 20)
    refs = [B.synth_ref_host(3, x, y - x + 3), B.synth_ref_host(4, x2, y2 - x2 + 3)]
    good, per = caller.blocks_records([(tpl, seq, x, y), (tpl2, seq2, x2, y2)], refs)
    assert len(good) == int(per.sum()) and per[0] > 0 and per[1] > 0
    bad = tpl2.copy()
    bad["orientation"][3] = 7
    with pytest.raises(B.BscError, match="template %d has orientation" % (len(tpl) + 3)):  # named by its index among the call's templates
        caller.blocks_records([(tpl, seq, x, y), (bad, seq2, x2, y2)], refs)
    left = tpl2.copy()
    left["pos"][0] = (x2 - 1, 0)
    with pytest.raises(B.BscError, match="left of the block start %d" % x2):  # ... and with its own block's start
        caller.blocks_records([(tpl, seq, x, y), (left, seq2, x2, y2)], refs)
    with pytest.raises(B.BscError, match="y \\("):
        caller.blocks_records([(tpl, seq, x, y), (tpl2, seq2, x2 + 5, x2)], [refs[0], np.zeros(0, dtype=np.uint8)])  # y < x
    small = np.zeros(10, dtype=B.VCF_REC)
    with pytest.raises(B.BscError, match="out_cap"):
        caller.blocks_records([(tpl, seq, x, y), (tpl2, seq2, x2, y2)], refs, out=small)
    # one call in flight per context, shared with the single-block entries
    out = np.zeros(len(good) + 8, dtype=B.VCF_REC)
    caller.blocks_records([(tpl, seq, x, y), (tpl2, seq2, x2, y2)], refs, out=out, submit_only=True)
    with pytest.raises(B.BscError, match="fetched"):
        caller.block_records(tpl, seq, x, y, refs[0])
    with pytest.raises(B.BscError, match="no block was submitted"):
        caller.block_records_fetch()
    got, per2 = caller.blocks_records_fetch()
    assert got.tobytes() == good.tobytes() and (per2 == per).all()
    assert len(caller.block_records(tpl, seq, x, y, refs[0])) == int(per[0])  # the context goes on


def test_blocks_submit_to_equals_block_by_block(caller, oracle, tables, libm_exact):
    """bsc_blocks_submit_to / bsc_blocks_submit_to_inplace (the gt_meth / gt_vcf form the drop-in glue batches small blocks through):
    every block's images are the bytes of bsc_call_block on that block alone AND of the CPU oracle (HOT LOOP A, then the calc
    threads' loop: oracle.accumulate + oracle.call_sites), the positions between blocks are images of nothing, counters add up."""
    rng = np.random.default_rng(77)
    blocks = [b[:4] + (b[4],) for b in _random_blocks(rng, 60, 1, 9_000, split_every=4)]
    refs = [b[4] for b in blocks]
    blocks = [b[:4] for b in blocks]
    caller.reset_stats()
    single = [caller.call_block(t, s, x, y, refs[i][: y - x + 1]) for i, (t, s, x, y) in enumerate(blocks)]
    st1 = caller.stats()
    # the path the drop-in runs, against the oracle itself (not only against another HIP path): every third block
    flav = oracle.LIBM if libm_exact else oracle.BSM
    for i in range(0, len(blocks), 3):
        t, s, x, y = blocks[i]
        rc, pile = oracle.accumulate(t, s, x, y, 20)
        assert rc == 0
        exp, eskip = oracle.call_sites(pile, refs[i][: y - x + 1], tables, flav, -8)
        assert single[i][0].tobytes() == exp.tobytes() and (single[i][1] == eskip).all(), i
    caller.reset_stats()
    for stride, inplace in ((200, False), (208, False), (208, True), (200, True)):
        off, out, skip = caller.blocks_submit_to(blocks, refs, out_stride=stride, inplace=inplace)
        for i, (t, s, x, y) in enumerate(blocks):
            n, o = y - x + 1, int(off[i])
            assert o % 64 == 0
            img = out[o : o + n]
            if stride == 208:
                assert img[:, :200].tobytes() == single[i][0].tobytes() and (img[:, 201] == single[i][1]).all() and not img[:, 200].any()
            else:
                assert img.tobytes() == single[i][0].tobytes(), i
            assert (skip[o : o + n] == single[i][1]).all()
            pad = ((n + 63) // 64) * 64 - n
            assert (skip[o + n : o + n + pad] == 1).all()
    st2 = caller.stats()
    assert st2["sites"] == 4 * st1["sites"] and st2["covered"] == 4 * st1["covered"] and st2["het_calls"] == 4 * st1["het_calls"]
    bad = blocks[3][0].copy()
    bad["bs_strand"][0] = 9
    with pytest.raises(B.BscError, match="bs_strand 9"):
        caller.blocks_submit_to(blocks[:3] + [(bad,) + blocks[3][1:]] + blocks[4:], refs)
    with pytest.raises(B.BscError, match="bs_strand 9"):
        caller.blocks_submit_to(blocks[:3] + [(bad,) + blocks[3][1:]] + blocks[4:], refs, inplace=True)
    assert caller.blocks_submit_to(blocks[:2], refs[:2], inplace=True)[1][: blocks[0][3] - blocks[0][2] + 1].tobytes() == single[0][0].tobytes()


def test_blocks_bcf_is_the_blocks_streams_one_after_another(caller):
    """bsc_blocks_bcf_submit[_inplace] / _fetch: several blocks in ONE launch sequence, ONE BCF stream back = the streams of bsc_block_bcf called on
    the blocks one after another (itself pinned to the host encoder over the oracle chain's records, tests/test_gpu_bcf.py): random blocks incl.
    adjacent ones cut at a CpG; sizes around the tile geometry with dbSNP flags, a names table over all the blocks, every position written, a
    region; a room that is too small answered by bsc_block_bcf_again; and the fetches do not take each other's submissions."""
    from bs_call_amd.caller import BscError

    rng = np.random.default_rng(77)
    raw = _random_blocks(rng, 120, 50, 9000, split_every=7)
    blocks = [(t, s, x, y) for t, s, x, y, _ in raw]
    refs = [r for _, _, _, _, r in raw]
    want = [caller.block_bcf(t, s, x, y, refs[i], 5) for i, (t, s, x, y) in enumerate(blocks)]
    exp, n_exp = b"".join(w[0] for w in want), sum(w[1] for w in want)
    for inplace in (False, True):
        got, n = caller.blocks_bcf(blocks, refs, 5, inplace=inplace)
        assert n == n_exp and got == exp, inplace
    # tile-geometry sizes, dbSNP flags and names
    sizes = [1, 2, 59, 60, 61, 62, 63, 64, 65, 127, 128, 129, 40_000, 7, 3000]
    blocks, pos = [], 500
    for i, n in enumerate(sizes):
        tpl, seq, x, y = _block(SEED + 41_000 + i, pos + 150, n, 30)
        blocks.append((tpl, seq, x, y))
        pos = y
    blocks.append((blocks[0][0][:0], blocks[0][1], pos + 10, pos + 300))  # a block without reads
    refs = [B.synth_ref_host(SEED + 41_000 + i, x, y - x + 3) for i, (_, _, x, y) in enumerate(blocks)]
    dbs = [rng.choice([0, 1, 3], size=y - x + 1, p=[0.9, 0.05, 0.05]).astype(np.uint8) for _, _, x, y in blocks]
    flagged = np.concatenate([np.flatnonzero(d) + x for d, (_, _, x, _) in zip(dbs, blocks)]).astype(np.uint32)
    listed = flagged[rng.random(len(flagged)) < 0.8]
    nm = [b"rs%d" % int(v) for v in rng.integers(1, 10**9, len(listed))]
    off = np.concatenate([[0], np.cumsum([len(v) for v in nm])]).astype(np.uint32)
    names = (listed, off, b"".join(nm))
    for kw in (dict(), dict(all_positions=True), dict(reg_start=blocks[5][2], reg_stop=blocks[-3][3] - 1000)):
        want = [caller.block_bcf(t, s, x, y, refs[i], 2, names=names, dbsnp=dbs[i], **kw) for i, (t, s, x, y) in enumerate(blocks)]
        got, n = caller.blocks_bcf(blocks, refs, 2, names=names, dbsnp=dbs, **kw)
        assert n == sum(w[1] for w in want) and got == b"".join(w[0] for w in want), kw
    exp = b"".join(w[0] for w in want)
    # too little room: refused with the length needed, then the encoder alone once more (the wrapper does that when cap is its own)
    with pytest.raises(BscError, match="stream has"):
        caller.blocks_bcf(blocks, refs, 2, names=names, dbsnp=dbs, cap=len(exp) // 2, **kw)
    got, n = caller.blocks_bcf(blocks, refs, 2, names=names, dbsnp=dbs, **kw)
    assert got == exp
    # a records fetch does not take a BCF submission, and the other way round
    import ctypes as C

    from bs_call_amd import _lib

    L = _lib.load()
    a, b = C.c_uint64(0), C.c_uint64(0)
    assert L.bsc_blocks_bcf_fetch(caller._h, C.byref(a), C.byref(b)) == -1
    caller.blocks_records(blocks, refs, dbsnp=dbs, submit_only=True)
    assert L.bsc_blocks_bcf_fetch(caller._h, C.byref(a), C.byref(b)) == -1 and b"bsc_blocks_bcf_submit" in L.bsc_last_error()
    caller.blocks_records_fetch()
