"""GPU parity of the accumulate stage (reference HOT LOOP A, src/call_genotypes.c:178-226) and of whole blocks
(accumulate + call), through the C ABI, against the CPU oracle: every byte of every pile-up / gt_meth record."""
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


def _block(seed, x0, n, cov, pad=2):
    tpl, seq = B.synth_reads_host(seed, x0, n, cov)
    x = max(1, x0 - pad)  # the reference starts the block 2 positions left of the first read (process_template.c:24-28)
    y = int(max((tpl["pos"] + tpl["len"]).max(), x0)) - 1 if len(tpl) else x0
    return tpl, seq, x, y


@pytest.mark.parametrize("cov,n,x0", [(10, 100_000, 1000), (30, 200_000, 5_000_000), (200, 30_000, 77), (2, 50_000, 3)])
def test_accumulate_parity(caller, oracle, cov, n, x0):
    tpl, seq, x, y = _block(SEED + cov, x0, n, cov)
    rc, exp = oracle.accumulate(tpl, seq, x, y, 20)
    assert rc == 0
    got = caller.accumulate(tpl, seq, x, y)
    assert got.tobytes() == exp.tobytes()
    assert abs(got["n"][100:-400].mean() - cov * 0.935) < cov * 0.05  # 3 % low quality, 1 % N, trimmed ends
    # the generator hands over whether read 0 "was walked" (bsc_template.flags); without the flags the device finds out itself
    from bs_call_amd.reads import walk_flags
    assert (tpl["flags"] == walk_flags(tpl, seq)).all() and (tpl["flags"] & 2 == 0).any()
    bare = tpl.copy()
    bare["flags"] = 0
    assert caller.accumulate(bare, seq, x, y).tobytes() == exp.tobytes()


@pytest.mark.parametrize("y_cut", [0, 1, 63, 64, 65, 150, 1000])
def test_window_clipping_and_ragged_tiles(caller, oracle, y_cut):
    """y inside the reads (pos <= y clipping, :214) and block sizes around the 64-position wave tile."""
    tpl, seq, x, y = _block(SEED, 10_000, 3_000, 30)
    y2 = x + y_cut
    keep = np.minimum(np.where(tpl["pos"][:, 0] > 0, tpl["pos"][:, 0], tpl["pos"][:, 1]),
                      np.where(tpl["pos"][:, 1] > 0, tpl["pos"][:, 1], tpl["pos"][:, 0])) <= y2 + 500
    rc, exp = oracle.accumulate(tpl[keep], seq, x, y2, 20)
    assert rc == 0
    got = caller.accumulate(tpl[keep], seq, x, y2)
    assert len(got) == y_cut + 1 and got.tobytes() == exp.tobytes()


def test_empty_single_end_and_unsorted(caller, oracle):
    # no templates at all
    got = caller.accumulate(np.zeros(0, dtype=B.TEMPLATE), np.zeros(0, dtype=np.uint8), 100, 300)
    assert len(got) == 201 and not got.tobytes().strip(b"\0")
    tpl, seq, x, y = _block(SEED + 3, 500, 20_000, 30)
    rc, exp = oracle.accumulate(tpl, seq, x, y, 20)
    # the sums do not depend on template order: a shuffled list gives the same pile-up
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(tpl))
    got = caller.accumulate(tpl[perm], seq, x, y)
    assert got.tobytes() == exp.tobytes()
    # reverse-read-only templates and the "skipped read 0 does not flip the orientation" quirk (:204,224)
    t2 = tpl.copy()
    sel = np.zeros(len(t2), dtype=bool)
    sel[::3] = True
    sel &= t2["len"][:, 1] > 0  # keep at least one read per template
    t2["len"][sel, 0] = 0
    t2["pos"][sel, 0] = 0
    seq2 = seq.copy()
    for t in t2[1::7]:  # read 0 entirely trimmed: q = 63 everywhere
        if t["len"][0]:
            o, l = int(t["off"][0]), int(t["len"][0])
            seq2[o : o + l] = (seq2[o : o + l] & 3) | (63 << 2)
    rc, exp2 = oracle.accumulate(t2, seq2, x, y, 20)
    assert rc == 0
    # the reads were edited after the generator flagged them: without the host's word (flags = 0), and with it made again
    from bs_call_amd.reads import walk_flags
    t2["flags"] = 0
    got2 = caller.accumulate(t2, seq2, x, y)
    assert got2.tobytes() == exp2.tobytes()
    t2["flags"] = walk_flags(t2, seq2)
    assert (t2["flags"] == 1).sum() > 100 and caller.accumulate(t2, seq2, x, y).tobytes() == exp2.tobytes()
    assert got2.tobytes() != exp.tobytes()


def test_min_qual_parameter(oracle):
    tpl, seq, x, y = _block(SEED + 5, 2_000, 20_000, 30)
    for mq in (1, 10, 30, 43):
        rc, exp = oracle.accumulate(tpl, seq, x, y, mq)
        with B.SiteCaller(min_qual=mq) as c:
            got = c.accumulate(tpl, seq, x, y)
        assert got.tobytes() == exp.tobytes(), mq


def test_reference_asserts_become_errors(caller):
    tpl, seq, x, y = _block(SEED, 1_000, 2_000, 10)
    with pytest.raises(B.BscError) as e:  # assert(y >= x), :158
        caller.accumulate(tpl, seq, 500, 499)
    assert e.value.code == -1
    with pytest.raises(B.BscError):  # assert(x1 >= x), :186
        caller.accumulate(tpl, seq, int(tpl["pos"][0, 0]) + 1, y)
    bad = tpl.copy()
    bad["orientation"][5] = 2
    with pytest.raises(B.BscError):  # assert(ori < 2), :188
        caller.accumulate(bad, seq, x, y)
    bad = tpl.copy()
    bad["off"][7, 1] = len(seq)
    with pytest.raises(B.BscError):
        caller.accumulate(bad, seq, x, y)


def test_inexact_range_is_reported(caller, oracle):
    """MAPQ 255 at depth > 258: the float sum of MAPQ^2 leaves the exact range; results are still produced and the
    call says so (BSC_WARN_INEXACT)."""
    n_t = 400
    tpl = np.zeros(n_t, dtype=B.TEMPLATE)
    tpl["pos"][:, 0] = 1000
    tpl["len"][:, 0] = 50
    tpl["off"][:, 0] = np.arange(n_t) * 50
    tpl["mapq"][:, 0] = 255
    tpl["bs_strand"] = 1
    seq = np.full(n_t * 50, 1 | (30 << 2), dtype=np.uint8)
    with pytest.warns(B.BscInexactWarning):
        got = caller.accumulate(tpl, seq, 998, 1060)
    assert got["n"][2] == n_t and got["counts"][2].sum() == n_t  # the integer fields are exact regardless
    # 258 reads stay exact and silent
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("error")
        got = caller.accumulate(tpl[:258], seq, 998, 1060)
    rc, exp = oracle.accumulate(tpl[:258], seq, 998, 1060, 20)
    assert got.tobytes() == exp.tobytes()


def test_call_block_parity(caller, oracle, tables, libm_exact):
    """One whole call_genotypes_ML block: reads in, gt_meth out, pile-up never on the host."""
    for cov, n, x0 in ((30, 150_000, 20_000), (300, 8_000, 64)):
        tpl, seq, x, y = _block(SEED + 11 + cov, x0, n, cov)
        ref = B.synth_ref_host(SEED + 11 + cov, x, y - x + 1)
        rc, pile = oracle.accumulate(tpl, seq, x, y, 20)
        exp, eskip = oracle.call_sites(pile, ref, tables, oracle.LIBM if libm_exact else oracle.BSM, -8)
        got, skip = caller.call_block(tpl, seq, x, y, ref)
        assert (skip == eskip).all() and got.tobytes() == exp.tobytes()
        raw, skip2 = caller.call_block(tpl, seq, x, y, ref, out_stride=208)
        assert raw[:, :200].tobytes() == exp.tobytes() and (raw[:, 201] == eskip).all()


def test_async_block_submit_fetch(caller, oracle, tables, libm_exact):
    """bsc_block_submit / bsc_block_fetch: the inputs are staged at submit (the caller may scribble over them right
    away, as bs_call recycles its align_list), one block in flight, results as bsc_call_block."""
    flav = oracle.LIBM if libm_exact else oracle.BSM
    blocks = []
    for i, (cov, n, x0) in enumerate(((30, 60_000, 5_000), (10, 100_000, 900_000), (30, 777, 2_000_000))):
        tpl, seq, x, y = _block(SEED + 40 + i, x0, n, cov)
        ref = B.synth_ref_host(SEED + 40 + i, x, y - x + 1)
        rc, pile = oracle.accumulate(tpl, seq, x, y, 20)
        exp, eskip = oracle.call_sites(pile, ref, tables, flav, -8)
        blocks.append((tpl, seq, x, y, ref, exp, eskip))
    for tpl, seq, x, y, ref, exp, eskip in blocks:
        t2, s2, r2 = tpl.copy(), seq.copy(), ref.copy()
        caller.block_submit(t2, s2, x, y, r2)
        t2["pos"][:] = 0  # recycle the inputs while the block is in flight
        s2[:] = 0
        r2[:] = 0
        with pytest.raises(B.BscError):  # one block in flight
            caller.block_submit(tpl, seq, x, y, ref)
        got, skip = caller.block_fetch()
        assert (skip == eskip).all() and got.tobytes() == exp.tobytes()
    with pytest.raises(B.BscError):
        caller._pending = (1, 200)
        caller.block_fetch()
    # destination named at submit (pinned): the copy-out rides behind the kernels, the fetch only waits and reports
    tpl, seq, x, y, ref, exp, eskip = blocks[0]
    p_out, p_skip = B.PinnedBuffer(y - x + 1, B.GT_METH), B.PinnedBuffer(y - x + 1, np.uint8)
    p_out.array.view(np.uint8)[:] = 0xEE
    caller.block_submit_to(tpl, seq, x, y, ref, p_out.array, p_skip.array)
    got, skip = caller.block_fetch()
    assert got is p_out.array and (skip == eskip).all() and got.tobytes() == exp.tobytes()
    raw = B.PinnedBuffer((y - x + 1, 208), np.uint8)
    caller.block_submit_to(tpl, seq, x, y, ref, raw.array, p_skip.array)
    got, skip = caller.block_fetch()
    assert got[:, :200].tobytes() == exp.tobytes() and (got[:, 201] == eskip).all()
    bad = tpl.copy()
    bad["orientation"][3] = 9
    caller.block_submit_to(bad, seq, x, y, ref, p_out.array, p_skip.array)
    with pytest.raises(B.BscError):
        caller.block_fetch()
    caller.block_submit_to(tpl, seq, x, y, ref, p_out.array, p_skip.array)  # the context is usable afterwards
    got, skip = caller.block_fetch()
    assert got.tobytes() == exp.tobytes()


def test_device_side_validation(caller, oracle, tables, libm_exact):
    """The templates are checked by the kernel that reads them: the first invalid template (lowest index, checks in
    the reference's order) is named, nothing is written for a bad block, a submitted block reports at fetch, and an
    unsorted submitted block is sorted and re-run behind the scenes."""
    tpl, seq, x, y = _block(SEED + 70, 10_000, 30_000, 20)
    ref = B.synth_ref_host(SEED + 70, x, y - x + 1)
    bad = tpl.copy()
    bad["orientation"][900] = 3
    bad["bs_strand"][900] = 7  # same template: the orientation check comes first
    bad["off"][400, 0] = len(seq) - 10  # read 0 of an earlier template runs past the buffer
    bad["bs_strand"][2500] = 3
    with pytest.raises(B.BscError) as e:
        caller.accumulate(bad, seq, x, y)
    assert "read 0 of template 400" in str(e.value)
    bad["off"][400, 0] = tpl["off"][400, 0]
    with pytest.raises(B.BscError) as e:
        caller.accumulate(bad, seq, x, y)
    assert "template 900 has orientation 3" in str(e.value)
    bad["orientation"][900] = 1
    with pytest.raises(B.BscError) as e:
        caller.accumulate(bad, seq, x, y)
    assert "template 900 has bs_strand 7" in str(e.value)
    # junk in the flags word (a caller built when it was padding): refused — the device trusts the two bits that are defined
    junk = tpl.copy()
    junk["flags"][77] |= 0x100
    with pytest.raises(B.BscError, match="template 77 has flags 0x1"):
        caller.accumulate(junk, seq, x, y)
    # nothing is written for a bad block
    out = np.full(y - x + 1, 0x5A, dtype=np.uint8).repeat(200).view(B.GT_METH)
    skip = np.full(y - x + 1, 0x5A, dtype=np.uint8)
    with pytest.raises(B.BscError):
        caller.call_block(bad, seq, x, y, ref, out=out, skip=skip)
    assert (out.view(np.uint8) == 0x5A).all() and (skip == 0x5A).all()
    # asynchronous form: the verdict arrives with the fetch, and the context stays usable
    caller.block_submit(bad, seq, x, y, ref)
    with pytest.raises(B.BscError) as e:
        caller.block_fetch()
    assert "template 900" in str(e.value)
    # an unsorted list through submit / fetch
    flav = oracle.LIBM if libm_exact else oracle.BSM
    rc, pile = oracle.accumulate(tpl, seq, x, y, 20)
    exp, eskip = oracle.call_sites(pile, ref, tables, flav, -8)
    perm = np.random.default_rng(5).permutation(len(tpl))
    caller.block_submit(tpl[perm], seq, x, y, ref)
    got, skip = caller.block_fetch()
    assert (skip == eskip).all() and got.tobytes() == exp.tobytes()
    got, skip = caller.call_block(tpl[perm], seq, x, y, ref)
    assert (skip == eskip).all() and got.tobytes() == exp.tobytes()


def test_plain_c_host_program():
    """integration/demo_block.c: a gcc-built C program drives the library through the C ABI alone."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bs_call_amd", "lib", "demo_block")
    assert os.path.exists(exe), "run `make demo`"
    first = None
    # the glue's protocol holds blocks back until a batch is worth a launch sequence (integration/amd_overlap_protocol.h): all
    # four blocks in one batch (default threshold, 1 M positions), two per batch, every block its own batch — same bytes
    for batch in (None, "90000", "0"):
        env = dict(os.environ)
        if batch is not None:
            env["BSCALL_AMD_BATCH_POSITIONS"] = batch
        r = subprocess.run([exe, "50000", "30", "4"], capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0, r.stderr + r.stdout
        ov = [ln for ln in r.stdout.strip().splitlines() if "overlapped" in ln][0].split("hash ")[1]
        first = first or ov
        assert ov == first
    lines = r.stdout.strip().splitlines()
    last = lines[-1]
    assert "positions called" in last and "VCF records" in last
    assert any(ln.startswith("chrS\t") for ln in lines)
    # the overlapped form of the replacement call_genotypes_ML (two pinned gt_vcf arrays, submit / fetch, a consumer thread
    # draining work->vcf[] on the ready flags like src/process.c:87-104) delivered the synchronous form's bytes
    ov = [ln for ln in lines if "overlapped" in ln]
    assert ov and "4 blocks" in ov[0]
    h = ov[0].split("hash ")[1].split(" / ")
    assert h[0].strip() == h[1].strip()
    # ... through the protocol code of integration/call_genotypes_amd_overlap.c itself, with a meth profiling thread reading
    # work->ref1 while the process thread overwrites it for the next block the moment each call returns
    mp = [ln for ln in lines if "profiling jobs" in ln]
    assert mp and " 0 found it changed" in mp[0] and "as handed over" in mp[0]


def test_prepared_templates_on_the_device(caller, oracle):
    """Reads with soft clips, indels and overlapping mates through bsc_prepare_templates (host C, src/process_template.c:
    36-111 restated) and then through the device accumulate: same pile-up as the oracle's accumulate over the same
    prepared templates."""
    from bs_call_amd.caller import prepare_templates
    from tests.test_prep import _random_read, to_arrays, tpl

    rng = np.random.default_rng(31)
    ts = []
    for i in range(3000):
        r0, m0, s0 = _random_read(rng, 15)
        r1, m1, s1 = _random_read(rng, 15)
        p0 = 1000 + 11 * i
        ts.append(tpl((p0, p0 + int(rng.integers(0, s0 + 40))), (s0, s1), (r0, r1), (m0, m1), orientation=int(rng.integers(0, 2)),
                      bs_strand=int(rng.integers(0, 3))))
    raw, seq, ms = to_arrays(ts)
    try:
        out, oseq, st = prepare_templates(raw, seq, ms)
    except B.BscError:
        pytest.skip("the random draw hit the reference's undefined indel-beyond-the-read class")
    x = 998
    y = int(max((int(o["pos"][k]) + int(o["len"][k])) for o in out for k in range(2) if o["len"][k])) + 1
    rc, exp = oracle.accumulate(out, oseq, x, y, 20)
    assert rc == 0 and int(st["base_overlap"]) > 0 and int(st["base_clip"]) > 0
    got = caller.accumulate(out, oseq, x, y)
    assert got.tobytes() == exp.tobytes()


def test_large_block_multi_chunk_copy_out(caller, oracle, tables, libm_exact):
    """A block longer than the 1 Mi-position copy-out chunk of bsc_call_block (3 chunks, last one ragged)."""
    tpl, seq, x, y = _block(SEED + 91, 3_000_000, 2_300_000, 10)
    ref = B.synth_ref_host(SEED + 91, x, y - x + 1)
    rc, pile = oracle.accumulate(tpl, seq, x, y, 20)
    assert rc == 0
    exp, eskip = oracle.call_sites(pile, ref, tables, oracle.LIBM if libm_exact else oracle.BSM, -8)
    got, skip = caller.call_block(tpl, seq, x, y, ref)
    assert (skip == eskip).all() and got.tobytes() == exp.tobytes()
    caller.block_submit(tpl, seq, x, y, ref, out_stride=208)
    raw, skip2 = caller.block_fetch()
    assert raw[:, :200].tobytes() == exp.tobytes() and (skip2 == eskip).all()


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_parameters_and_blocks(oracle, seed, libm_exact):
    """Random model parameters x random small blocks (coverage, length, start, min_qual)."""
    rng = np.random.default_rng(1000 + seed)
    under, over = float(rng.uniform(0, 0.3)), float(rng.uniform(0, 0.3))
    rb = float(rng.choice([0.5, 1.0, 2.0, 7.5]))
    mq = int(rng.integers(1, 44))
    cov = int(rng.choice([1, 3, 15, 40, 120]))
    n = int(rng.integers(100, 40_000))
    x0 = int(rng.integers(3, 50_000_000))
    tb = oracle.Tables(under, over, rb, mq)
    tpl, seq, x, y = _block(SEED + seed, x0, n, cov)
    ref = B.synth_ref_host(SEED + seed, x, y - x + 1, flags=1 if seed % 2 else 0)
    rc, pile = oracle.accumulate(tpl, seq, x, y, mq)
    exp, eskip = oracle.call_sites(pile, ref, tb, oracle.LIBM if libm_exact else oracle.BSM, -8)
    with B.SiteCaller(under, over, rb, mq) as c:
        assert c.accumulate(tpl, seq, x, y).tobytes() == pile.tobytes()
        got, skip = c.call_block(tpl, seq, x, y, ref)
        assert (skip == eskip).all() and got.tobytes() == exp.tobytes()
        rec = c.vcf_records(got, skip, np.concatenate([ref, [1, 2]]).astype(np.uint8), x)
    if libm_exact:
        assert rec.tobytes() == oracle.vcf_block(exp, eskip, np.concatenate([ref, [1, 2]]).astype(np.uint8), x).tobytes()


@pytest.mark.parametrize("seed", range(5))
def test_adversarial_template_lists(caller, oracle, seed):
    """Hand-rolled template lists that no read simulator produces: arbitrary order, mates far apart or swapped, reads of
    1..400 bases, reads hanging over the block end or starting beyond it, empty and fully trimmed reads, a pile of
    identical templates, and blocks from one to a few hundred tiles — the ordering, the tile search (largest template
    extent) and the window clipping of the device stage against the reference's plain loop."""
    rng = np.random.default_rng(4242 + seed)
    n_pos = int(rng.choice([1, 63, 64, 65, 1000, 20_000]))
    x = int(rng.integers(1, 1_000_000))
    y = x + n_pos - 1
    nt = int(rng.choice([0, 1, 7, 300, 5000]))
    if seed == 4:
        n_pos, nt = 20_000, 5000
        y = x + n_pos - 1
    tpl = np.zeros(nt, dtype=B.TEMPLATE)
    chunks, off = [], 0
    for i in range(nt):
        fwd = x + int(rng.integers(0, n_pos + 50))  # some start right of the block
        gap = int(rng.choice([0, 1, 50, 300, 3000])) * int(rng.choice([1, -1]))
        rev = max(x, fwd + gap)
        tpl["orientation"][i] = rng.integers(0, 2)
        tpl["bs_strand"][i] = rng.integers(0, 3)
        for k, pos in ((0, fwd), (1, rev)):
            kind = rng.integers(0, 10)
            if kind == 0:
                continue  # read absent: pos 0, len 0
            rl = int(rng.choice([1, 2, 30, 100, 400]))
            q = rng.integers(1, 44, size=rl).astype(np.uint8)
            b = rng.integers(0, 4, size=rl).astype(np.uint8)
            if kind == 1:
                q[:] = 63  # fully trimmed
            elif kind == 2:
                q[: rl // 3] = 63
                q[rl - rl // 4:] = 0
            tpl["pos"][i, k] = pos
            tpl["len"][i, k] = rl
            tpl["off"][i, k] = off
            tpl["mapq"][i, k] = rng.integers(0, 61)
            chunks.append(b | (q << 2))
            off += rl
        if tpl["len"][i, 0] == 0 and tpl["len"][i, 1] == 0:  # the reference never sees a template without reads
            tpl["pos"][i, 0], tpl["len"][i, 0], tpl["off"][i, 0] = x, 1, off
            chunks.append(np.array([1 | (30 << 2)], dtype=np.uint8))
            off += 1
    if nt >= 300:  # a pile of identical templates and a hot spot
        tpl[10:60] = tpl[5]
    seq = np.concatenate(chunks) if chunks else np.zeros(1, dtype=np.uint8)
    rc, exp = oracle.accumulate(tpl, seq, x, y, 20)
    assert rc == 0
    got = caller.accumulate(tpl, seq, x, y)
    assert got.tobytes() == exp.tobytes()
    perm = rng.permutation(nt)
    assert caller.accumulate(tpl[perm], seq, x, y).tobytes() == exp.tobytes()
    # the same list with the host's word on read 0 (fully trimmed / quality-0 read 0s are among them: :203,:210,:224)
    from bs_call_amd.reads import walk_flags
    told = tpl.copy()
    told["flags"] = walk_flags(tpl, seq)
    assert caller.accumulate(told, seq, x, y).tobytes() == exp.tobytes()


def test_far_apart_mates_do_not_widen_the_search(caller, oracle):
    """One template whose mates lie 2.9 Mb apart among 30 000 ordinary ones: the tiles search reads, not templates, so
    the candidate window stays one read long (ordered by template extent, every tile would walk the whole block)."""
    import time

    tpl, seq = B.synth_reads_host(SEED + 77, 1000, 3_000_000, 2)
    x, y = 998, int((tpl["pos"] + tpl["len"]).max()) - 1
    far = tpl[:1].copy()
    far["pos"][0] = [1000, y - 150]
    far["len"][0] = [100, 100]
    far["off"][0] = [tpl["off"][0, 0], tpl["off"][0, 1]]  # reuse read bytes that exist
    far["flags"] = 0  # not the generator's template any more: let the device look at read 0
    tpl2 = np.concatenate([tpl, far])
    rc, exp = oracle.accumulate(tpl2, seq, x, y, 20)
    assert rc == 0
    caller.accumulate(tpl2, seq, x, y)  # warm-up (allocations)
    t0 = time.perf_counter()
    got = caller.accumulate(tpl2, seq, x, y)
    dt = time.perf_counter() - t0
    assert got.tobytes() == exp.tobytes()
    assert dt < 0.5, "accumulate took %.2f s" % dt
