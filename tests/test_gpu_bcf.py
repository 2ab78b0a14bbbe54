"""The BCF encoder on the device (csrc/bcfdev.hip: bsc_bcf_block_device, bsc_block_bcf, bsc_block_bcf_raw) against the host
encoder (csrc/bcf.c, itself pinned to the independent BCF2 encoder / reader of oracle/py_bcf.py by tests/test_bcf.py) and against
py_bcf directly: random packed records incl. every size class of every typed value, records that are not written, names from a
table, tiles whose stream does not fit the wave's image (two passes), capacities that are too small, refused records; then whole
blocks — reads in, BCF bytes out — against bsc_block_records[_raw] + bsc_bcf_block, with a dbSNP index naming records."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import bs_call_amd as B
from bs_call_amd import _lib, vcf
from bs_call_amd.abi import VCF_REC
from oracle import py_bcf

import oracle_chain as OC
import test_bcf as TB

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def caller():
    c = B.SiteCaller()
    yield c
    c.close()


def _host_stream(recs, rid, names=None, ids=None):
    """bsc_bcf_record over the records, names from the table (rs_found set and position listed), at most 63 bytes of a name"""
    L = _lib.load()
    if ids is None:
        ids = _lib.BcfIds()
        L.bsc_bcf_default_ids(C.byref(ids))
    table = {}
    if names is not None:
        pos, off, by = names
        table = {int(p): by[int(off[i]) : int(off[i + 1])][:63] for i, p in enumerate(pos)}
    buf = (C.c_uint8 * 512)()
    out = []
    for r in recs:
        rs = table.get(int(r["core"]["pos"]), b"") if int(r["rs_found"]) else b""
        r1 = np.ascontiguousarray(r).reshape(1)
        n = L.bsc_bcf_record(r1.ctypes.data, rid, rs, len(rs), C.byref(ids), buf, 512)
        assert 0 <= n <= 512
        out.append(bytes(buf[:n]))
    return out


def _device_stream(caller, recs, rid, names=None, ids=None, cap=None, n_recs=None, max_recs=None):
    import torch

    dev = torch.device("cuda:0")
    raw = np.ascontiguousarray(recs).view(np.uint8).reshape(-1)
    d_recs = torch.from_numpy(raw.copy()).to(dev) if len(raw) else torch.zeros(128, dtype=torch.uint8, device=dev)
    n = len(recs) if n_recs is None else n_recs
    d_n = torch.tensor([n], dtype=torch.int64, device=dev)
    cap = 336 * max(len(recs), 1) if cap is None else cap
    d_out = torch.full((max(cap, 1) + 64,), 0xEE, dtype=torch.uint8, device=dev)
    d_tot = torch.zeros(3, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    caller.bcf_block_device(d_recs.data_ptr(), d_n.data_ptr(), len(recs) if max_recs is None else max_recs, rid, d_out.data_ptr(), cap,
                            d_tot.data_ptr(), names=names, ids=ids, stream=st)
    torch.cuda.synchronize()
    tot = d_tot.cpu().numpy()
    out = d_out.cpu().numpy()
    assert (out[cap:] == 0xEE).all(), "bytes behind the capacity were written"
    return out[:cap], int(tot[0]), int(tot[1])


def _random_block(rng, n, named=0.0, long_names=False):
    recs = np.zeros(n, dtype=VCF_REC)
    pos = np.sort(rng.choice(np.arange(1, 50 * n + 100), n, replace=False)).astype(np.uint32)
    for i in range(n):
        recs[i] = TB._random_rec(rng)
        recs[i]["core"]["pos"] = pos[i]
        if rng.random() < 0.12:
            recs[i]["core"]["emit"] = 0
        if rng.random() < named:
            recs[i]["rs_found"] = int(rng.choice([1, 3]))
        if rng.random() < 0.1:
            recs[i]["core"]["fs"] = int(rng.choice([-5, 127, 128, 32767, 32768, 2_000_000]))
            recs[i]["mq"] = int(rng.choice([0, 127, 128, 40000]))
            recs[i]["core"]["qd"] = int(rng.choice([0, 127, 128, 32768]))
        if rng.random() < 0.05:
            recs[i]["counts"] = 0  # no AMQ field
        if rng.random() < 0.03:
            recs[i]["counts"][int(rng.integers(0, 8))] = 0xFFFFFFFF  # the saturated count: -1 as the encoder's int32
    # the table lists most flagged positions, a few that no record has, and leaves some flagged records without a name
    listed = [int(p) for p, r in zip(pos, recs) if r["rs_found"] and rng.random() < 0.85]
    listed += [int(p) + 1 for p in pos[:: max(1, n // 7)] if int(p) + 1 not in set(pos.tolist())]
    listed = sorted(set(listed))
    names_b, off = [], [0]
    for p in listed:
        if long_names:
            nm = b"rs" + bytes(rng.choice(list(b"0123456789"), int(rng.integers(10, 90))).tolist())
        else:
            nm = b"rs%d" % int(rng.integers(1, 10**9))
            if len(nm) % 2 == 1 and rng.random() < 0.5:
                nm += b"\0"  # an odd number of digits carries its filler (bsc_dbsnp_name)
        names_b.append(nm)
        off.append(off[-1] + len(nm))
    names = (np.array(listed, dtype=np.uint32), np.array(off, dtype=np.uint32), b"".join(names_b))
    return recs, names


def test_random_records_device_equals_host_and_python(caller):
    rng = np.random.default_rng(20261004)
    for trial, n in enumerate((1, 63, 64, 65, 700, 4097)):
        recs, names = _random_block(rng, n, named=0.3)
        rid = trial % 25
        want = _host_stream(recs, rid, names)
        got, total, bad = _device_stream(caller, recs, rid, names)
        exp = b"".join(want)
        assert total == len(exp) and bad == 0
        assert got[:total].tobytes() == exp, (n, "device stream differs from bsc_bcf_record's")
        # the independent Python encoder on a sample (the host form is pinned to it record by record in tests/test_bcf.py)
        table = {int(p): names[2][int(names[1][i]) : int(names[1][i + 1])] for i, p in enumerate(names[0])}
        for j in range(0, n, max(1, n // 40)):
            r = recs[j]
            if not r["core"]["emit"]:
                assert want[j] == b""
                continue
            rs = table.get(int(r["core"]["pos"]), b"") if r["rs_found"] else b""
            assert want[j] == py_bcf.encode_record(TB._as_dict(r), rid, rs)
            dec = py_bcf.decode_record(want[j])
            assert dec["pos"] == int(r["core"]["pos"]) and dec["id"] == rs
    # without a table no record is named, whatever its flag says
    recs, names = _random_block(rng, 300, named=0.5)
    got, total, bad = _device_stream(caller, recs, 3, None)
    assert got[:total].tobytes() == b"".join(_host_stream(recs, 3, None))


def test_long_names_and_wide_dictionary_indices_take_the_two_pass_path(caller):
    """64 records of > 168 bytes each do not fit the wave's image (8 192 bytes): the tile goes out in two halves"""
    rng = np.random.default_rng(77)
    recs, names = _random_block(rng, 1500, named=0.9, long_names=True)
    recs["core"]["emit"] = 1
    ids = _lib.BcfIds(*[int(v) for v in rng.choice([5, 127, 128, 300, 32767, 32768, 1_000_000], 17)])
    want = _host_stream(recs, 24, names, ids)
    sizes = np.array([len(w) for w in want])
    assert sizes.max() > 230 and sizes[:64].sum() > 10752
    got, total, bad = _device_stream(caller, recs, 24, names, ids)
    assert total == sizes.sum() and bad == 0 and got[:total].tobytes() == b"".join(want)


def test_tiles_of_the_longest_records_go_out_in_four_parts(caller):
    """the wave's image is 8 KB: 32 records of ~310 bytes do not fit it, 16 do — four parts of 16 lanes (eight parts exist for smaller images
    only: 16 x 336 < 8 192).  Every size class at its widest: 63-byte names, four-byte dictionary indices, counts beyond 32 767, all four
    filter names, six likelihoods, two ALT alleles, a heterozygous call's FS; a stretch of short records in between, so that the parts of
    one tile differ in what they hold.  (The per-position form's tiles take the same code.)"""
    rng = np.random.default_rng(4242)
    recs, _ = _random_block(rng, 900, named=1.0, long_names=True)
    recs["core"]["emit"] = 1
    recs["rs_found"] = 1
    recs["core"]["gt"] = 1          # heterozygous: FS is written
    recs["core"]["gt_enc"] = 0x24
    recs["core"]["flt"] = 15        # q20;qd2;fs60;mq40
    recs["core"]["n_gl"] = 6
    recs["core"]["alt"] = b"CT"
    recs["core"]["fs"] = 2_000_000
    recs["core"]["qd"] = 40000
    recs["core"]["dp"] = 100000
    recs["mq"] = 40000
    recs["counts"] = 70000 + np.arange(8, dtype=np.uint32)
    recs["qual"] = 43
    recs["counts"][200:230] = 5
    recs["core"]["n_gl"][200:230] = 1
    listed = np.sort(recs["core"]["pos"]).astype(np.uint32)  # every record named: the longest name the encoder keeps, and longer ones (cut at 63)
    nm = [b"rs" + bytes(rng.choice(list(b"0123456789"), int(rng.integers(61, 90))).tolist()) for _ in listed]
    off = np.concatenate([[0], np.cumsum([len(x) for x in nm])]).astype(np.uint32)
    names = (listed, off, b"".join(nm))
    ids = _lib.BcfIds(*[1_000_000 + k for k in range(17)])
    want = _host_stream(recs, 24, names, ids)
    sizes = np.array([len(w) for w in want])
    assert sizes.max() <= 336 and sizes[:32].sum() > 8192 and sizes[:16].sum() <= 8192
    got, total, bad = _device_stream(caller, recs, 24, names, ids)
    assert total == sizes.sum() and bad == 0 and got[:total].tobytes() == b"".join(want)


def test_counts_capacities_and_refused_records(caller):
    rng = np.random.default_rng(5)
    recs, names = _random_block(rng, 1000, named=0.2)
    want = _host_stream(recs, 1, names)
    exp = b"".join(want)
    # the count on the device decides, not the array's size; nothing is written for n = 0
    got, total, _ = _device_stream(caller, recs, 1, names, n_recs=333)
    assert got[:total].tobytes() == b"".join(want[:333])
    got, total, _ = _device_stream(caller, recs, 1, names, n_recs=5000)  # more than the array holds: clamped
    assert got[:total].tobytes() == exp
    got, total, _ = _device_stream(caller, recs, 1, names, n_recs=0)
    assert total == 0
    got, total, _ = _device_stream(caller, recs[:0], 1, None, max_recs=0)
    assert total == 0
    # a capacity that is too small: the full length is reported, what was written is a prefix cut at a 64-record boundary
    cap = len(exp) // 2
    got, total, _ = _device_stream(caller, recs, 1, names, cap=cap)
    assert total == len(exp)
    ends = np.cumsum([sum(len(w) for w in want[k : k + 64]) for k in range(0, 1000, 64)])
    whole = int(ends[ends <= cap].max())
    assert got[:whole].tobytes() == exp[:whole] and (got[whole:] == 0xEE).all()
    # records bsc_bcf_record refuses are counted
    bad = recs.copy()
    bad["core"]["gt"][10] = 12
    bad["core"]["n_gl"][500] = 7
    bad["core"]["emit"][[10, 500]] = 1
    _, _, n_bad = _device_stream(caller, bad, 1, names)
    assert n_bad == 2
    # argument checks
    with pytest.raises(B.BscError):
        _device_stream(caller, recs, 1, (names[0][::-1].copy(), names[1], names[2]))  # positions must ascend


def _reads_block(seed, x, n, cov):
    from bs_call_amd.reads import synth_block

    tpl, seq, y = synth_block(seed, x, n, cov)
    ref = B.synth_ref_host(seed, x, y - x + 3)
    return tpl, seq, y, ref


def test_block_bcf_equals_block_records_then_the_host_encoder(caller):
    x, n = 20_000, 150_000
    tpl, seq, y, ref = _reads_block(88172645463325252 + 41, x, n, 30)
    caller.reset_site_stats()
    recs = caller.block_records(tpl, seq, x, y, ref, with_stats=True).copy()
    st_want = caller.site_stats().copy()
    want = vcf.bcf_block(recs, 7)
    caller.reset_site_stats()
    for rep in range(3):  # the first call sizes its copy-out after the wait, the later ones ahead of it
        got, n_rec = caller.block_bcf(tpl, seq, x, y, ref, 7, with_stats=(rep == 0))
        assert n_rec == len(recs) > 50_000 and got == want
    # ... and, with no product code in between: the CPU oracle chain's records through the Python encoder
    core, gtm = OC.records(tpl, seq, x, y, ref)
    OC.same_records(recs, core, gtm)
    assert got == OC.bcf_stream(core, gtm, 7)
    from tests.test_gpu_chain import _same_stats

    _same_stats(caller.site_stats().copy(), st_want)  # (the two methylation profiles are float sums of atomics: equal within 1e-12)
    # every record decodes; positions ascend
    o, last, k = 0, 0, 0
    while o < len(got):
        ln = 8 + int.from_bytes(got[o : o + 4], "little") + int.from_bytes(got[o + 4 : o + 8], "little")
        if k % 5000 == 0:
            d = py_bcf.decode_record(got[o : o + ln])
            assert d["pos"] > last and d["rid"] == 7
            last = d["pos"]
        o += ln
        k += 1
    assert o == len(got) and k == n_rec
    # -A: a record for every position
    recs_all = caller.block_records(tpl, seq, x, y, ref, all_positions=True)
    got, n_rec = caller.block_bcf(tpl, seq, x, y, ref, 0, all_positions=True)
    assert n_rec == len(recs_all) >= y - x and got == vcf.bcf_block(recs_all, 0)
    core, gtm = OC.records(tpl, seq, x, y, ref, all_positions=True)
    OC.same_records(recs_all, core, gtm)
    head = OC.bcf_stream(core[:20_000], gtm[:20_000], 0)
    assert got[: len(head)] == head
    # out_cap too small: an error that names the size needed, and the context goes on
    with pytest.raises(B.BscError, match="bytes, out_cap is"):
        caller.block_bcf(tpl, seq, x, y, ref, 7, cap=len(want) - 1)
    got, _ = caller.block_bcf(tpl, seq, x, y, ref, 7, cap=len(want))
    assert got == want
    # an empty block
    got, n_rec = caller.block_bcf(tpl[:0], seq[:0], x, x + 99, ref[:102], 7)
    assert n_rec == 0 and got == b""


def test_block_bcf_split_form_and_two_contexts_alternating(caller):
    """bsc_block_bcf_submit / _fetch: the bytes of the blocking call; one block in flight per context, whichever entry queued it; two
    contexts alternating from one thread (block k + 1 queued while block k's stream is still on its way back) write the same streams."""
    blocks = []
    for k in range(4):
        x = 50_000 + 90_000 * k
        tpl, seq, y, ref = _reads_block(88172645463325252 + 60 + k, x, 60_000 + 7_001 * k, 30)
        blocks.append((tpl, seq, x, y, ref))
    want = [caller.block_bcf(tpl, seq, x, y, ref, k) for k, (tpl, seq, x, y, ref) in enumerate(blocks)]
    tpl, seq, x, y, ref = blocks[0]
    out = B.PinnedBuffer(200 * (y - x + 1), np.uint8)
    with pytest.raises(B.BscError, match="no block was submitted"):
        caller.block_bcf_fetch()
    caller.block_bcf_submit(tpl, seq, x, y, ref, 0, out.array)
    tpl[:] = 0  # the inputs were staged: the caller's arrays are free at once
    with pytest.raises(B.BscError, match="fetched"):
        caller.block_bcf_submit(*blocks[1][:5], 1, out.array)
    with pytest.raises(B.BscError, match="fetched"):
        caller.block_records(blocks[1][0], blocks[1][1], blocks[1][2], blocks[1][3], blocks[1][4])
    with pytest.raises(B.BscError, match="no block was submitted"):
        caller.block_records_fetch()
    got, n_rec = caller.block_bcf_fetch()
    assert (got.tobytes(), n_rec) == want[0]
    out.free()
    blocks[0] = _reads_block(88172645463325252 + 60, 50_000, 60_000, 30)
    blocks[0] = (blocks[0][0], blocks[0][1], 50_000, blocks[0][2], blocks[0][3])
    with B.SiteCaller() as c2:
        ctxs = [caller, c2]
        outs = [B.PinnedBuffer(200 * 100_000, np.uint8), B.PinnedBuffer(200 * 100_000, np.uint8)]
        got = []
        for k, (tpl, seq, x, y, ref) in enumerate(blocks):
            c = ctxs[k & 1]
            if k >= 2:
                g, n = c.block_bcf_fetch()
                got.append((g.tobytes(), n))
            c.block_bcf_submit(tpl, seq, x, y, ref, k, outs[k & 1].array, inplace=bool(k & 2))  # staged and in place in turn
        for k in (2, 3):
            g, n = ctxs[k & 1].block_bcf_fetch()
            got.append((g.tobytes(), n))
        for o in outs:
            o.free()
    assert got == want


def test_sites_form_equals_packing_then_encoding(caller):
    """bsc_bcf_sites_device: the per-position arrays the reads-in chain leaves (records + the packed half) -> the stream, no packing pass;
    the bytes of bsc_vcf_compact_device + bsc_bcf_block_device and of the host encoder over the packed records, with names, for block
    lengths around the 64-position tiles."""
    import torch

    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(99)
    for x, n_sites in ((7_000, 64), (7_000, 1_000), (123_457, 70_001)):
        tpl, seq, y, ref = _reads_block(88172645463325252 + 47 + n_sites, x, n_sites, 25)
        n = y - x + 1
        flags = (rng.random(n) < 0.02).astype(np.uint8) * rng.choice(np.array([1, 3], dtype=np.uint8), n)
        listed = np.flatnonzero(flags)[::2] + x  # every second flagged position has a name
        nm = [b"rs%d" % int(v) for v in rng.integers(1, 10**8, len(listed))]
        off = np.concatenate([[0], np.cumsum([len(b) for b in nm])]).astype(np.uint32)
        names = (listed.astype(np.uint32), off, b"".join(nm))
        d_tpl = torch.from_numpy(tpl.view(np.uint8).reshape(-1).copy()).to(dev)
        d_seq = torch.from_numpy(seq).to(dev)
        d_ref = torch.from_numpy(ref).to(dev)
        d_db = torch.from_numpy(flags).to(dev)
        d_core = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        d_aux = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        caller.reads_chain_device(d_tpl.data_ptr(), len(tpl), d_seq.data_ptr(), seq.size, x, y, d_ref.data_ptr(), d_core.data_ptr(), d_aux=d_aux.data_ptr(),
                                  d_dbsnp=d_db.data_ptr(), stream=st)
        caller.block_status(st)
        d_rec = torch.empty(n * 128, dtype=torch.uint8, device=dev)
        d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        caller.vcf_compact_device(d_core.data_ptr(), d_aux.data_ptr(), 0, n, d_rec.data_ptr(), n, d_cnt.data_ptr(), stream=st)
        n_rec = int(d_cnt.item())
        recs = d_rec[: n_rec * 128].cpu().numpy().view(VCF_REC)
        want = b"".join(_host_stream(recs, 9, names))
        cap = len(want) + 100
        d_out = torch.full((cap + 64,), 0xEE, dtype=torch.uint8, device=dev)
        d_tot = torch.zeros(3, dtype=torch.int64, device=dev)
        caller.bcf_sites_device(d_core.data_ptr(), d_aux.data_ptr(), n, 9, d_out.data_ptr(), cap, d_tot.data_ptr(), names=names, stream=st)
        torch.cuda.synchronize()
        tot = d_tot.cpu().numpy()
        out = d_out.cpu().numpy()
        assert int(tot[0]) == len(want) and int(tot[1]) == 0 and int(tot[2]) == n_rec > 0, (n_sites, tot, n_rec)
        assert out[: len(want)].tobytes() == want and (out[len(want) :] == 0xEE).all(), n_sites
        assert int((recs["rs_found"] != 0).sum()) > 0 or n_sites < 1000
        got2, total2, _ = _device_stream(caller, recs, 9, names)
        assert got2[:total2].tobytes() == want
        # ... and the CPU oracle chain's records (with the same dbSNP flags) through the Python encoder, named from the same list
        core, gtm = OC.records(tpl, seq, x, y, ref, dbsnp=flags)
        OC.same_records(recs, core, gtm)
        name_at = {int(p): b for p, b in zip(listed, nm)}
        assert want == OC.bcf_stream(core, gtm, 9, lambda pos: name_at.get(pos) if int(flags[pos - x]) else None)


def test_block_bcf_names_its_records_from_a_dbsnp_index(caller, tmp_path):
    from bs_call_amd.dbsnp import DbSnpIndex

    spec = importlib.util.spec_from_file_location("make_dbsnp_index", os.path.join(ROOT, "tools", "make_dbsnp_index.py"))
    W = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(W)
    x, n = 5_000, 120_000
    tpl, seq, y, ref = _reads_block(88172645463325252 + 43, x, n, 20)
    path = str(tmp_path / "names.idx")
    sites = W.synthetic_sites(x + n + 500, 40)
    W.write_index(path, {"ctgN": sites})
    with DbSnpIndex(path) as db:
        db.load_contig("ctgN")
        flags = db.flags(x, y - x + 1)
        names = db.names(x, y - x + 1)
        # the table is bsc_dbsnp_name's answer for every flagged position of the range
        assert len(names[0]) == int((flags != 0).sum()) > 1000
        for i in range(0, len(names[0]), 97):
            r, nm, ln = db.name(int(names[0][i]))
            assert r and names[2][int(names[1][i]) : int(names[1][i + 1])] == (nm.encode() + b"\0")[:ln]
        recs = caller.block_records(tpl, seq, x, y, ref, dbsnp=flags)
        want = vcf.bcf_block(recs, 2, db)
        got, n_rec = caller.block_bcf(tpl, seq, x, y, ref, 2, names=names, dbsnp=flags)
    assert n_rec == len(recs) and int((recs["rs_found"] != 0).sum()) > 500
    assert got == want
    assert b"rs" in got


def test_split_form_with_names_takes_the_table_with_the_uploads(caller, tmp_path):
    """bsc_block_bcf_submit_inplace with a names table in ordinary memory: the table is the caller's again when the call returns (it is
    scribbled over before the fetch), the stream is the blocking call's, and the call returns long before the block is through (the table
    goes up with the block's other inputs, ahead of its kernels — queued behind them from pageable memory it held the call back)."""
    import time

    rng = np.random.default_rng(5)
    x, n = 9_000, 1_500_000
    tpl, seq, y, ref = _reads_block(88172645463325252 + 71, x, n, 30)
    sz = y - x + 1
    flags = (rng.random(sz) < 0.03).astype(np.uint8) * 3
    listed = (np.flatnonzero(flags) + x).astype(np.uint32)
    nm = [b"rs%d" % int(v) for v in rng.integers(1, 10**9, len(listed))]
    off = np.concatenate([[0], np.cumsum([len(b) for b in nm])]).astype(np.uint32)
    names = (listed, off, b"".join(nm))
    want, n_want = caller.block_bcf(tpl, seq, x, y, ref, 5, names=names, dbsnp=flags)
    assert n_want > 500_000 and want.count(b"rs") > 10_000
    ids = _lib.BcfIds()
    caller._L.bsc_bcf_default_ids(C.byref(ids))
    p = _lib.VcfParams(0, 1, 0xFFFFFFFF)
    pin = {k: B.PinnedBuffer(max(1, a.nbytes), np.uint8) for k, a in (("tpl", tpl), ("seq", seq), ("ref", ref), ("db", flags))}
    for k, a in (("tpl", tpl), ("seq", seq), ("ref", ref), ("db", flags)):
        pin[k].array[: a.nbytes] = a.view(np.uint8).reshape(-1)
    out = B.PinnedBuffer(len(want) + 4096, np.uint8)
    ratios = []
    for rep in range(3):
        pos_a, off_a, by_a = listed.copy(), off.copy(), np.frombuffer(names[2] + b"\0", dtype=np.uint8).copy()
        st = _lib.BcfNames(pos_a.ctypes.data, off_a.ctypes.data, by_a.ctypes.data, len(pos_a))
        t0 = time.perf_counter()
        rc = caller._L.bsc_block_bcf_submit_inplace(caller._h, pin["tpl"].array.ctypes.data, len(tpl), pin["seq"].array.ctypes.data, seq.size, x, y,
                                                    pin["ref"].array.ctypes.data, pin["db"].array.ctypes.data, C.byref(p), 0, 5, C.byref(ids), C.addressof(st),
                                                    out.array.ctypes.data, out.array.size)
        t1 = time.perf_counter()
        assert rc == 0, rc
        pos_a[:] = 0  # the table was copied by the call
        off_a[:] = 0
        by_a[:] = 0x41
        nb, nr = C.c_uint64(0), C.c_uint64(0)
        assert caller._L.bsc_block_bcf_fetch(caller._h, C.byref(nb), C.byref(nr)) == 0
        t2 = time.perf_counter()
        assert (out.array[: nb.value].tobytes(), nr.value) == (want, n_want)
        ratios.append((t1 - t0) / (t2 - t0))
    assert min(ratios[1:]) < 0.5, ratios  # (the first pass sizes workspaces)
    for b in list(pin.values()) + [out]:
        b.free()


def test_stream_longer_than_the_room_is_encoded_again_not_computed_again(caller):
    """bsc_block_bcf_again: a block refused for its out_cap is encoded once more from what it left in HBM — the bytes of a call with enough
    room, and bsc_get_stats / the site statistics count the block ONCE (round 5's callers ran the whole block a second time)."""
    x, n = 30_000, 120_000
    tpl, seq, y, ref = _reads_block(88172645463325252 + 73, x, n, 30)
    caller.reset_site_stats()
    s0 = caller.stats()
    want, n_want = caller.block_bcf(tpl, seq, x, y, ref, 6, with_stats=True)
    s1 = caller.stats()
    st_want = caller.site_stats().copy()
    caller.reset_site_stats()
    ids = _lib.BcfIds()
    caller._L.bsc_bcf_default_ids(C.byref(ids))
    p = _lib.VcfParams(0, 1, 0xFFFFFFFF)
    small = np.empty(len(want) // 3, dtype=np.uint8)
    nb, nr = C.c_uint64(0), C.c_uint64(0)
    t8, s8, r8 = np.ascontiguousarray(tpl), np.ascontiguousarray(seq), np.ascontiguousarray(ref)
    rc = caller._L.bsc_block_bcf(caller._h, t8.ctypes.data, len(t8), s8.ctypes.data, s8.size, x, y, r8.ctypes.data, None, C.byref(p), 1, 6, C.byref(ids), None,
                                 small.ctypes.data, small.size, C.byref(nb), C.byref(nr))
    assert rc == -1 and nb.value == len(want)
    too_small = np.empty(len(want) - 1, dtype=np.uint8)
    assert caller._L.bsc_block_bcf_again(caller._h, too_small.ctypes.data, too_small.size, C.byref(nb), C.byref(nr)) == -1 and nb.value == len(want)
    big = np.empty(len(want) + 100, dtype=np.uint8)
    assert caller._L.bsc_block_bcf_again(caller._h, big.ctypes.data, big.size, C.byref(nb), C.byref(nr)) == 0
    assert (big[: nb.value].tobytes(), nr.value) == (want, n_want)
    s2 = caller.stats()
    for k in ("sites", "covered", "het_calls"):
        assert s2[k] - s1[k] == s1[k] - s0[k] > 0, k
    from tests.test_gpu_chain import _same_stats

    _same_stats(caller.site_stats().copy(), st_want)
    # nothing to encode again now; and not after another kind of block either
    assert caller._L.bsc_block_bcf_again(caller._h, big.ctypes.data, big.size, C.byref(nb), C.byref(nr)) == -1
    # the Python mirror takes that path by itself when the default room is too small (-A over a block: ~190 bytes per position would fit; force it)
    got, n_got = caller.block_bcf(tpl, seq, x, y, ref, 6)
    assert (got, n_got) == (want, n_want)


def test_block_bcf_raw_equals_block_records_raw_then_the_host_encoder(caller):
    import test_prep as T
    from oracle import py_prep

    rng = np.random.default_rng(1234)
    ts, p0 = [], 3000
    for i in range(4000):
        r0, m0, s0 = T._random_read(rng, 20)
        r1, m1, s1 = T._random_read(rng, 20)
        p0 += int(rng.integers(0, 5))
        t = T.tpl((p0, p0 + int(rng.integers(0, s0 + 40))), (s0, s1), (r0, r1), (m0, m1))
        t["orientation"] = int(rng.integers(0, 2))
        t["bs_strand"] = int(rng.integers(1, 3))
        try:
            py_prep.prepare([t])
            ts.append(t)
        except py_prep.PrepError:
            pass
    raw, seq, ms = T.to_arrays(ts)
    kw = dict(left_trim=(1, 0), right_trim=(0, 2), min_qual=20)
    x = max(1, int(min(p for p in raw["pos"].ravel() if p)) - 2)
    y = int((raw["pos"].astype(np.int64) + raw["reference_span"]).max()) + 60
    ref = B.synth_ref_host(11, x, y - x + 3)
    caller.reset_site_stats()
    recs, st = caller.block_records_raw(raw, seq, ms, x, y, ref, with_stats=True, **kw)
    stats = caller.site_stats().copy()
    want = vcf.bcf_block(recs, 4)
    caller.reset_site_stats()
    got, n_rec, st2 = caller.block_bcf_raw(raw, seq, ms, x, y, ref, 4, with_stats=True, **kw)
    assert n_rec == len(recs) > 1000 and got == want and st2.tobytes() == st.tobytes()
    # ... and, with no product code in between: py_prep -> the oracle's accumulate / call / record formation -> the Python encoder
    o_tpl, o_seq, o_st = OC.prepare(ts, **kw)
    assert {f: int(st2[f]) for f in st2.dtype.names} == o_st
    core, gtm = OC.records(o_tpl, o_seq, x, y, ref)
    OC.same_records(recs, core, gtm)
    assert got == OC.bcf_stream(core, gtm, 4)
    from tests.test_gpu_chain import _same_stats

    _same_stats(caller.site_stats().copy(), stats)


def test_encoder_over_a_whole_contig_of_records(caller):
    """8 M positions at 30x, everything resident in HBM (pile-up -> calling kernel -> record formation -> packing -> encoder): the stream
    of ALL records equals the host encoder's; walking it from record to record by the two lengths in front of each ends exactly at its
    end, positions ascending; every 50 000th record decodes (independent reader) to the packed record's fields."""
    import torch

    n, x0, cov = 8_000_000, 1_000, 30
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    d_cts = torch.empty((n + 2) * 104, dtype=torch.uint8, device=dev)
    d_ref = torch.empty(n + 2, dtype=torch.uint8, device=dev)
    d_out = torch.empty(n * 200, dtype=torch.uint8, device=dev)
    d_skip = torch.empty(n, dtype=torch.uint8, device=dev)
    d_vcf = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    d_rec = torch.empty(n * 128, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    d_tot = torch.zeros(3, dtype=torch.int64, device=dev)
    caller.synth_device(88172645463325252 + 77, x0, n + 2, cov, d_cts.data_ptr(), d_ref.data_ptr(), 0, st)
    caller.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out.data_ptr(), d_skip.data_ptr(), 200, st)
    caller.vcf_records_device(d_out.data_ptr(), 200, d_skip.data_ptr(), d_ref.data_ptr(), n, x0, d_vcf.data_ptr(), stream=st)
    caller.vcf_compact_device(d_vcf.data_ptr(), d_out.data_ptr(), 200, n, d_rec.data_ptr(), n, d_cnt.data_ptr(), stream=st)
    n_rec = int(d_cnt.item())
    assert 0.4 * n < n_rec < 0.7 * n
    cap = n_rec * 140
    d_bcf = torch.empty(cap, dtype=torch.uint8, device=dev)
    caller.bcf_block_device(d_rec.data_ptr(), d_cnt.data_ptr(), n, 5, d_bcf.data_ptr(), cap, d_tot.data_ptr(), stream=st)
    torch.cuda.synchronize()
    total, bad = int(d_tot[0].item()), int(d_tot[1].item())
    assert bad == 0 and total <= cap
    got = d_bcf[:total].cpu().numpy()
    recs = d_rec[: n_rec * 128].cpu().numpy().view(VCF_REC)
    del d_cts, d_out, d_vcf, d_rec, d_bcf
    want = vcf.bcf_block(recs, 5)
    assert len(want) == total and got.tobytes() == want
    # the walk
    buf = got.tobytes()
    o, k, last = 0, 0, 0
    while o < total:
        ln = 8 + int.from_bytes(buf[o : o + 4], "little") + int.from_bytes(buf[o + 4 : o + 8], "little")
        if k % 50_000 == 0:
            d = py_bcf.decode_record(buf[o : o + ln])
            r = recs[k]
            assert d["pos"] == int(r["core"]["pos"]) > last and d["rid"] == 5 and d["fmt"]["MC8"] == [int(v) for v in r["counts"]]
            assert d["fmt"]["DP"] == [int(r["core"]["dp"])] and d["fmt"]["MQ"] == [int(r["mq"])] and d["qual"] == float(r["core"]["phred"])
            last = d["pos"]
        o += ln
        k += 1
    assert o == total and k == n_rec


def test_length_bytes_entries_and_a_length_that_lies(caller):
    """The device-level pair behind the block entries (bsc_reads_chain_len_device, bsc_bcf_sites_len_device: the chain's byte per position gates and
    sizes the encoder) writes the stream of bsc_bcf_sites_device (sizes from the records) — plain, with every position written (-A), and with wide
    dictionary indices (the bytes are then a gate only) — and a byte that does NOT hold its record's length makes the block fail: the record is
    counted with the refused ones (a stream with a hole or an overlap is never handed out silently)."""
    import torch
    from bs_call_amd import reads as R

    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    x, n_pos, cov = 5000, 300_000, 25
    tpl, seq, y = R.synth_block(77, x, n_pos, cov)
    ref = B.synth_ref_host(77, x, y - x + 3)
    n = y - x + 1
    up = lambda v: torch.from_numpy(v.view(np.uint8).reshape(-1).copy()).to(dev)
    d_tpl, d_seq, d_ref = up(tpl), up(seq), up(ref)
    for all_positions, ids in ((False, None), (True, None), (False, _lib.BcfIds(*[300 + k for k in range(17)]))):
        d_core = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        d_aux = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        d_len = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
        caller.reads_chain_len_device(d_tpl.data_ptr(), len(tpl), d_seq.data_ptr(), seq.size, x, y, d_ref.data_ptr(), d_core.data_ptr(), d_aux.data_ptr(),
                                      d_len.data_ptr(), all_positions=all_positions, stream=st)
        caller.block_status(st)
        cap = n * 130 + 4096
        outs = []
        for with_len in (False, True):
            d_out = torch.full((cap,), 0xEE, dtype=torch.uint8, device=dev)
            d_tot = torch.zeros(3, dtype=torch.int64, device=dev)
            if with_len:
                caller.bcf_sites_len_device(d_core.data_ptr(), d_aux.data_ptr(), d_len.data_ptr(), n, 3, d_out.data_ptr(), cap, d_tot.data_ptr(), ids=ids, stream=st)
            else:
                caller.bcf_sites_device(d_core.data_ptr(), d_aux.data_ptr(), n, 3, d_out.data_ptr(), cap, d_tot.data_ptr(), ids=ids, stream=st)
            torch.cuda.synchronize()
            tot = d_tot.cpu().numpy()
            assert tot[1] == 0 and tot[2] > (n // 3 if not all_positions else n * 9 // 10)
            outs.append((int(tot[0]), int(tot[2]), d_out[: int(tot[0])].cpu().numpy().tobytes()))
        assert outs[0] == outs[1], (all_positions, ids is not None)
        if ids is None and not all_positions:
            # a byte is not 0 exactly where a record is written
            core = d_core.cpu().numpy().view(B.VCF_CORE)
            lens = d_len[:n].cpu().numpy()
            assert ((lens != 0) == (core["emit"] != 0)).all()
            # one byte lies about its record by one: the block is refused
            k = int(np.flatnonzero((lens > 0) & (lens < 250))[1000])
            d_len[k] += 1
            d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
            d_tot = torch.zeros(3, dtype=torch.int64, device=dev)
            caller.bcf_sites_len_device(d_core.data_ptr(), d_aux.data_ptr(), d_len.data_ptr(), n, 3, d_out.data_ptr(), cap, d_tot.data_ptr(), stream=st)
            torch.cuda.synchronize()
            assert int(d_tot[1].item()) == 1 and int(d_tot[0].item()) == outs[1][0] + 1
