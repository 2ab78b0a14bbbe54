"""Argument checking of the C ABI on a live context: every entry refuses NULL / misaligned / inconsistent arguments with
BSC_ERR_ARG (-1) and a message, and the context stays usable afterwards."""
import ctypes as C

import numpy as np
import pytest

import bs_call_amd as B
from bs_call_amd import _lib

pytestmark = pytest.mark.gpu


def test_c_abi_rejects_bad_arguments():
    import torch

    L = _lib.load()
    with B.SiteCaller() as c:
        h = c._h
        n = 1000
        pile, ref = B.synth_pileup_host(3, 100, n + 2, 20)
        out = np.zeros(n, dtype=B.GT_METH)
        skip = np.zeros(n, dtype=np.uint8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        err = lambda: L.bsc_last_error().decode()
        # NULL pieces and a bad stride
        assert L.bsc_call_sites(h, None, p(ref), n, p(out), 200, p(skip)) == -1 and err()
        assert L.bsc_call_sites(h, p(pile), p(ref), n, p(out), 199, p(skip)) == -1 and "stride" in err()
        assert L.bsc_call_sites(None, p(pile), p(ref), n, p(out), 200, p(skip)) == -1
        assert L.bsc_call_sites(h, p(pile), p(ref), 0, p(out), 200, p(skip)) == 0  # an empty block is fine
        # device entries: misaligned pointers
        d = torch.zeros(n * 264 + 64, dtype=torch.uint8, device="cuda:0")
        base = d.data_ptr()
        assert base % 16 == 0
        st = None
        assert L.bsc_call_sites_device(h, base + 4, base, n, base, 200, base, st) == -1 and "align" in err()
        vp = _lib.VcfParams(0, 1, 0xFFFFFFFF)
        assert L.bsc_vcf_records_device(h, base + 4, 200, base, base, None, n, 1, C.byref(vp), base, st) == -1
        assert L.bsc_vcf_stats_device(h, base + 8, base, 200, None, n, st) == -1 and "align" in err()
        assert L.bsc_vcf_stats_device(h, base, base, 123, None, n, st) == -1
        assert L.bsc_vcf_compact_device(h, base, base, 200, None, n, base, n, None, st) == -1
        assert L.bsc_vcf_compact_device(h, base, base + 4, 200, None, n, base, n, base, st) == -1
        # blocks: y < x, NULL reference, fetch without submit, two submits
        tpl, seq = B.synth_reads_host(5, 1000, 2000, 10)
        x, y = 998, int((tpl["pos"] + tpl["len"]).max()) - 1
        ref2 = B.synth_ref_host(5, x, y - x + 3)
        cnt = C.c_uint64(0)
        rec = np.zeros(y - x + 1, dtype=B.VCF_REC)
        assert L.bsc_block_records(h, p(tpl), len(tpl), p(seq), len(seq), x, y, None, None, C.byref(vp), 0, p(rec), len(rec),
                                   C.byref(cnt)) == -1
        assert L.bsc_block_records(h, p(tpl), len(tpl), p(seq), len(seq), y, x, p(ref2), None, C.byref(vp), 0, p(rec), len(rec),
                                   C.byref(cnt)) == -1
        assert L.bsc_block_records(h, None, len(tpl), p(seq), len(seq), x, y, p(ref2), None, C.byref(vp), 0, p(rec), len(rec),
                                   C.byref(cnt)) == -1
        assert L.bsc_block_fetch(h, p(out), p(skip)) == -1 and "no block" in err()
        assert L.bsc_block_submit(h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(ref2), 200) == 0
        assert L.bsc_block_submit(h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(ref2), 200) == -1 and "fetched" in err()
        out2 = np.zeros(y - x + 1, dtype=B.GT_METH)
        skip2 = np.zeros(y - x + 1, dtype=np.uint8)
        assert L.bsc_block_fetch(h, None, None) == -1  # this block was submitted without a destination ...
        assert L.bsc_block_fetch(h, p(out2), p(skip2)) == 0  # ... and is still there for a proper fetch
        assert out2["counts"].sum() > 0
        # round 3's entries: device-resident reads, the reads-in chain, the split records form
        nrec = y - x + 1
        d_tpl = torch.from_numpy(tpl.view(np.uint8).reshape(-1)).to("cuda:0")
        d_seq = torch.from_numpy(seq).to("cuda:0")
        d_ref = torch.from_numpy(ref2).to("cuda:0")
        d_cts = torch.zeros(((nrec + 63) // 64 * 64) * 104, dtype=torch.uint8, device="cuda:0")
        d_core = torch.zeros(nrec * 64, dtype=torch.uint8, device="cuda:0")
        A = (h, d_tpl.data_ptr(), len(tpl), d_seq.data_ptr(), len(seq))
        assert L.bsc_accumulate_device(*A, x, y, None, st) == -1
        assert L.bsc_accumulate_device(*A, y, x, d_cts.data_ptr(), st) == -1 and "y (" in err()
        assert L.bsc_accumulate_device(*A, x, y, d_cts.data_ptr() + 4, st) == -1 and "align" in err()
        assert L.bsc_accumulate_device(h, None, len(tpl), d_seq.data_ptr(), len(seq), x, y, d_cts.data_ptr(), st) == -1
        assert L.bsc_reads_chain_device(*A, x, y, None, None, C.byref(vp), 0, d_core.data_ptr(), None, st) == -1
        assert L.bsc_reads_chain_device(*A, x, y, d_ref.data_ptr(), None, None, 0, d_core.data_ptr(), None, st) == -1
        assert L.bsc_reads_chain_device(*A, x, y, d_ref.data_ptr(), None, C.byref(vp), 0, d_core.data_ptr() + 8, None, st) == -1 and "align" in err()
        assert L.bsc_reads_chain_device(*A, y, x, d_ref.data_ptr(), None, C.byref(vp), 0, d_core.data_ptr(), None, st) == -1
        ms = C.c_float()
        assert L.bsc_last_reads_chain_ms(h, C.byref(ms)) == -1 and L.bsc_last_accumulate_ms(h, C.byref(ms)) == -1  # profiling is off
        assert L.bsc_accumulate_device(*A, x, y, d_cts.data_ptr(), st) == 0 and L.bsc_block_status(h, st) == 0
        assert L.bsc_reads_chain_device(*A, x, y, d_ref.data_ptr(), None, C.byref(vp), 0, d_core.data_ptr(), None, st) == 0
        assert L.bsc_block_status(h, st) == 0 and int(d_core.view(nrec, 64)[:, 4].sum()) > 0
        assert L.bsc_block_records_fetch(h, C.byref(cnt)) == -1 and "no block" in err()
        assert L.bsc_block_records_submit(h, p(tpl), len(tpl), p(seq), len(seq), x, y, None, None, C.byref(vp), 0, p(rec), len(rec)) == -1
        assert L.bsc_block_records_submit(h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(ref2), None, C.byref(vp), 0, p(rec), len(rec)) == 0
        assert L.bsc_block_records(h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(ref2), None, C.byref(vp), 0, p(rec), len(rec),
                                   C.byref(cnt)) == -1 and "fetched" in err()  # one block in flight per context
        assert L.bsc_block_records_fetch(h, None) == -1  # NULL count: the block stays pending ...
        assert L.bsc_block_records_fetch(h, C.byref(cnt)) == 0 and cnt.value > 0  # ... for a proper fetch
        small = np.zeros(8, dtype=B.VCF_REC)
        assert L.bsc_block_records_submit(h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(ref2), None, C.byref(vp), 0, p(small), len(small)) == 0
        need = C.c_uint64(0)
        assert L.bsc_block_records_fetch(h, C.byref(need)) == -1 and need.value == cnt.value and "out_cap" in err()
        # the in-place form: same argument checks, same one-block-in-flight rule, same fetch
        SI = L.bsc_block_records_submit_inplace
        assert SI(h, p(tpl), len(tpl), p(seq), len(seq), x, y, None, None, C.byref(vp), 0, p(rec), len(rec)) == -1
        assert SI(h, p(tpl), len(tpl), p(seq), len(seq), y, x, p(ref2), None, C.byref(vp), 0, p(rec), len(rec)) == -1
        assert SI(h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(ref2), None, C.byref(vp), 0, p(rec), len(rec)) == 0
        assert SI(h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(ref2), None, C.byref(vp), 0, p(rec), len(rec)) == -1 and "fetched" in err()
        n_in = C.c_uint64(0)
        assert L.bsc_block_records_fetch(h, C.byref(n_in)) == 0 and n_in.value == cnt.value
        # the two asynchronous block forms share the staging area, the device workspaces and the verdict counters: a block
        # submitted through either keeps every other host-buffer block entry out until it has been fetched
        pile_h = np.zeros(y - x + 1, dtype=B.PILEUP)
        assert L.bsc_block_records_submit(h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(ref2), None, C.byref(vp), 0, p(rec), len(rec)) == 0
        assert L.bsc_block_submit(h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(ref2), 200) == -1 and "fetched" in err()
        assert L.bsc_accumulate(h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(pile_h)) == -1 and "fetched" in err()
        assert L.bsc_call_block(h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(ref2), p(out2), 200, p(skip2)) == -1 and "fetched" in err()
        assert L.bsc_block_records_fetch(h, C.byref(n_in)) == 0 and n_in.value == cnt.value  # untouched by the refused calls
        assert rec[: cnt.value]["core"]["emit"].all()
        assert L.bsc_block_submit(h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(ref2), 200) == 0
        assert L.bsc_block_records(h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(ref2), None, C.byref(vp), 0, p(rec), len(rec),
                                   C.byref(cnt)) == -1 and "fetched" in err()
        assert L.bsc_block_records_submit(h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(ref2), None, C.byref(vp), 0, p(rec), len(rec)) == -1
        assert L.bsc_accumulate(h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(pile_h)) == -1 and "fetched" in err()
        out3 = np.zeros(y - x + 1, dtype=B.GT_METH)
        assert L.bsc_block_fetch(h, p(out3), p(skip2)) == 0 and out3.tobytes() == out2.tobytes()  # ... and neither was this one
        # round 5's entries: the device pre-processing, the raw-template block, the in-place gt_vcf form of several blocks
        from bs_call_amd.abi import BLOCK_DESC, MISMS, PREP_PARAMS, PREP_STATS, RAW_TEMPLATE

        raw = np.zeros(len(tpl), dtype=RAW_TEMPLATE)
        for f in ("pos", "len", "off", "mapq", "orientation", "bs_strand"):
            raw[f] = tpl[f]
        raw["reference_span"] = tpl["len"]
        par = np.zeros(1, dtype=PREP_PARAMS)
        par["min_qual"] = 20
        no_ms = np.zeros(1, dtype=MISMS)
        d_raw = torch.from_numpy(raw.view(np.uint8).reshape(-1)).to("cuda:0")
        d_po = torch.zeros(len(seq) + 64, dtype=torch.uint8, device="cuda:0")
        d_pt = torch.zeros(len(raw) * 40, dtype=torch.uint8, device="cuda:0")
        used, pst = C.c_uint64(0), np.zeros(1, dtype=PREP_STATS)
        PD = L.bsc_prepare_templates_device
        args_ok = (h, d_raw.data_ptr(), len(raw), d_seq.data_ptr(), len(seq), None, 0, p(par), d_pt.data_ptr(), d_po.data_ptr(), len(seq) + 64,
                   C.byref(used), p(pst), None, st)
        assert PD(*args_ok) == 0 and used.value == len(seq) and int(pst["reads"][0]) > 0
        assert PD(h, None, len(raw), *args_ok[3:]) == -1 and PD(h, d_raw.data_ptr() + 4, *args_ok[2:]) == -1 and "align" in err()
        assert PD(*args_ok[:7], None, *args_ok[8:]) == -1 and PD(*args_ok[:11], None, *args_ok[12:]) == -1
        assert PD(*args_ok[:10], 100, *args_ok[11:]) == -1 and "seq_out too small" in err()  # the output buffer cannot hold the reads
        bad = raw.copy()
        bad["orientation"][5] = 3
        d_bad = torch.from_numpy(bad.view(np.uint8).reshape(-1)).to("cuda:0")
        assert PD(h, d_bad.data_ptr(), *args_ok[2:]) == -1 and "template 5 has orientation 3" in err()
        bad = raw.copy()
        bad["off"][7, 1] = len(seq)
        d_bad = torch.from_numpy(bad.view(np.uint8).reshape(-1)).to("cuda:0")
        assert PD(h, d_bad.data_ptr(), *args_ok[2:]) == -1 and "read 1 of template 7 lies outside" in err()
        RAWF = L.bsc_block_records_raw
        r_ok = (h, p(raw), len(raw), p(seq), len(seq), p(no_ms), 0, p(par), x, y, p(ref2), None, C.byref(vp), 0, p(rec), len(rec), C.byref(cnt), p(pst), None)
        assert RAWF(*r_ok) == 0 and cnt.value == n_in.value
        assert RAWF(h, None, *r_ok[2:]) == -1 and RAWF(*r_ok[:7], None, *r_ok[8:]) == -1 and RAWF(*r_ok[:10], None, *r_ok[11:]) == -1
        assert RAWF(*r_ok[:8], y, x, *r_ok[10:]) == -1 and "y (" in err()
        desc = np.zeros(1, dtype=BLOCK_DESC)
        desc[0] = (x, y, len(tpl), 0)
        img, sk, off = B.PinnedBuffer((y - x + 64, 208), np.uint8), B.PinnedBuffer(y - x + 64, np.uint8), np.zeros(1, dtype=np.uint64)
        INP = L.bsc_blocks_submit_to_inplace
        assert INP(h, p(desc), 1, p(tpl), p(seq), len(seq), None, p(img.array), 208, p(sk.array), p(off)) == -1
        assert INP(h, p(desc), 0, p(tpl), p(seq), len(seq), p(ref2), p(img.array), 208, p(sk.array), p(off)) == -1 and "n_blocks" in err()
        assert INP(h, p(desc), 1, p(tpl), p(seq), len(seq), p(ref2), p(img.array), 123, p(sk.array), p(off)) == -1 and "stride" in err()
        assert INP(h, p(desc), 1, p(tpl), p(seq), len(seq), p(ref2), p(img.array), 208, p(sk.array), p(off)) == 0
        assert INP(h, p(desc), 1, p(tpl), p(seq), len(seq), p(ref2), p(img.array), 208, p(sk.array), p(off)) == -1 and "fetched" in err()
        assert L.bsc_block_fetch(h, None, None) == 0 and img.array[: y - x + 1, :200].tobytes() == out2.tobytes()
        img.free()
        sk.free()
        # round 5's BCF entries: the encoder on the device, from packed records, from the chain's arrays, as a whole block
        ids = _lib.BcfIds()
        L.bsc_bcf_default_ids(C.byref(ids))
        d_rec = torch.zeros(256 * 128 + 64, dtype=torch.uint8, device="cuda:0")
        d_cntb = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        d_bo = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda:0")
        BD, BS = L.bsc_bcf_block_device, L.bsc_bcf_sites_device
        b_ok = (h, d_rec.data_ptr(), d_cntb.data_ptr(), 256, 0, C.byref(ids), None, d_bo.data_ptr(), 1 << 16, d_cntb.data_ptr() + 8, st)
        assert BD(*b_ok) == 0 and L.bsc_synchronize(h) == 0  # zero records: an empty stream
        assert BD(h, None, *b_ok[2:]) == -1 and BD(*b_ok[:2], None, *b_ok[3:]) == -1 and BD(*b_ok[:5], None, *b_ok[6:]) == -1
        assert BD(*b_ok[:7], None, *b_ok[8:]) == -1 and BD(*b_ok[:9], None, st) == -1
        assert BD(h, d_rec.data_ptr() + 8, *b_ok[2:]) == -1 and "align" in err()
        assert BD(*b_ok[:2], d_cntb.data_ptr() + 4, *b_ok[3:]) == -1 and "align" in err()
        bad_names = _lib.BcfNames(None, None, None, 3)
        assert BD(*b_ok[:6], C.addressof(bad_names), *b_ok[7:]) == -1 and "names table" in err()
        s_ok = (h, d_core.data_ptr(), d_core.data_ptr(), 100, 0, C.byref(ids), None, d_bo.data_ptr(), 1 << 16, d_cntb.data_ptr() + 8, st)
        assert BS(h, None, *s_ok[2:]) == -1 and BS(*s_ok[:2], None, *s_ok[3:]) == -1 and BS(h, d_core.data_ptr() + 4, *s_ok[2:]) == -1
        nb, nrb = C.c_uint64(7), C.c_uint64(7)
        blob = np.zeros(200 * (y - x + 1), dtype=np.uint8)
        BB = L.bsc_block_bcf
        k_ok = (h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(ref2), None, C.byref(vp), 0, 1, C.byref(ids), None, p(blob), len(blob), C.byref(nb), C.byref(nrb))
        assert BB(*k_ok) == 0 and nrb.value == n_in.value and 90 * nrb.value < nb.value < 140 * nrb.value
        assert BB(*k_ok[:12], None, *k_ok[13:]) == -1 and BB(*k_ok[:14], None, *k_ok[15:]) == -1 and BB(*k_ok[:16], None, C.byref(nrb)) == -1
        assert BB(*k_ok[:7], None, *k_ok[8:]) == -1 and nb.value == 0 and nrb.value == 0  # a refused call leaves no stale sizes behind
        assert BB(*k_ok[:5], y, x, *k_ok[7:]) == -1 and "y (" in err()
        assert BB(*k_ok[:15], 1000, C.byref(nb), C.byref(nrb)) == -1 and "out_cap" in err() and nb.value > 1000  # too small: the size needed
        assert L.bsc_block_records_submit(h, p(tpl), len(tpl), p(seq), len(seq), x, y, p(ref2), None, C.byref(vp), 0, p(rec), len(rec)) == 0
        assert BB(*k_ok) == -1 and "fetched" in err()  # one block in flight per context
        assert L.bsc_block_records_fetch(h, C.byref(n_in)) == 0 and n_in.value == n_in.value
        assert BB(*k_ok) == 0 and nrb.value == n_in.value
        # after all that the context still computes
        got = c.block_records(tpl, seq, x, y, ref2)
        assert len(got) > 0 and (got["core"]["emit"] == 1).all()
        ss = np.zeros(1, dtype=B.SITE_STATS)
        assert L.bsc_get_site_stats(h, None) == -1 and L.bsc_get_site_stats(h, p(ss)) == 0


def test_a_fresh_contexts_first_block_sees_its_own_counters():
    """bsc_create clears the device counters with hipMemset, which returns before the bytes are written, and the context's stream does not
    wait for the null stream: before the create waited for it, the zeroes could land BEHIND the first block's all-ones "no error" word and
    the block was refused ("read -4 of template 0 lies outside the read buffer": the word read back as 0) — once in ~2 400 fresh contexts
    of tools/fuzz_block.py.  Many fresh contexts, one small block each, the first thing they do: none is refused, all give the same pile-up;
    the same for the statistics counters after bsc_reset_stats."""
    tpl, seq = B.synth_reads_host(11, 5000, 3000, 10)
    x, y = 4998, int((tpl["pos"] + tpl["len"]).max()) - 1
    first = None
    for k in range(150):
        with B.SiteCaller() as c:
            got = c.accumulate(tpl, seq, x, y).tobytes()
            first = got if first is None else first
            assert got == first, k
            if k % 10 == 0:
                c.reset_stats()
                assert c.accumulate(tpl, seq, x, y).tobytes() == first
                assert c.stats()["sites"] == 0
