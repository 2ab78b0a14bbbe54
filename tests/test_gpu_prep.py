"""Read pre-processing on the device (csrc/prepdev.hip, bsc_prepare_templates_device) against the host form
(csrc/prep.c) and the pure-Python restatement (oracle/py_prep.py): the hand-worked cases of tests/test_prep.py — one per branch of
trim_read / trim_soft_clips / handle_overlap / the indel normalisation, each derived by hand from the cited lines of the reference —
and random valid alignments, incl. the quirks (read_utils.c:22: the right trim copies the base of sp[k1]; al_utils.c: the overlap
walk over a shortened list).  Byte equality of everything that comes out: templates (positions, lengths, offsets, flags), read
bytes, statistics."""
import numpy as np
import pytest

import bs_call_amd as B
from bs_call_amd.caller import prepare_templates
from oracle import py_prep

import oracle_chain as OC
import test_prep as T  # the hand-worked cases and the generators live with the host form's tests

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def caller():
    c = B.SiteCaller()
    yield c
    c.close()


def _both(caller, templates, **kw):
    """device == host C == Python on one template list; returns the host result"""
    raw, seq, ms = T.to_arrays(templates)
    try:
        h_tpl, h_seq, h_st = prepare_templates(raw, seq, ms, **kw)
    except B.BscError as e:
        with pytest.raises(B.BscError) as d:  # the same refusal, in the host form's words (it is run on the offending template)
            caller.prepare_templates_device(raw, seq, ms, **kw)
        for key in ("soft clip", "indel beyond the read", "orientation", "outside the"):
            assert (key in str(e)) == (key in str(d.value)), (str(e), str(d.value))
        return None
    d_tpl, d_seq, d_st = caller.prepare_templates_device(raw, seq, ms, **kw)
    assert d_seq.tobytes() == h_seq.tobytes()
    for f in h_tpl.dtype.names:
        assert (d_tpl[f] == h_tpl[f]).all(), f
    assert d_tpl.tobytes() == h_tpl.tobytes() and d_st.tobytes() == h_st.tobytes()
    return h_tpl, h_seq, h_st


def test_hand_worked_cases_on_the_device(caller, monkeypatch):
    """Every hand-worked case of tests/test_prep.py once more, with the device form checked inside its `both` / `run_c`."""
    seen = []
    real_run_c = T.run_c

    def run_c(templates, **kw):
        got = real_run_c(templates, **kw)
        _both(caller, templates, **kw)
        seen.append(len(templates))
        return got

    monkeypatch.setattr(T, "run_c", run_c)
    for name in ("test_fixed_trims_and_the_right_trim_quirk", "test_soft_clips", "test_overlap_equal_spans_quality_decides_right_trim_of_read0",
                 "test_overlap_longer_span_wins_left_trim_moves_the_start", "test_overlap_reverse_read_first",
                 "test_no_overlap_and_single_reads_are_left_alone", "test_overlap_right_trim_walks_the_indels",
                 "test_overlap_left_trim_walks_the_indels", "test_indel_normalisation_alone"):
        getattr(T, name)()
    assert len(seen) >= 9


def test_random_alignments_device_equals_host(caller):
    rng = np.random.default_rng(20250505)
    n_err = 0
    for trial in range(250):
        ts = []
        for _ in range(int(rng.integers(1, 40))):
            r0, m0, s0 = T._random_read(rng)
            r1, m1, s1 = T._random_read(rng)
            kind = rng.random()
            p0 = int(rng.integers(50, 5000))
            if kind < 0.15:
                t = T.tpl((p0, 0), (s0, 0), (r0, None), (m0, ()))
            elif kind < 0.3:
                t = T.tpl((0, p0), (0, s1), (None, r1), ((), m1))
            else:
                p1 = p0 + int(rng.integers(-s1 - 5, s0 + 30))
                t = T.tpl((p0, max(1, p1)), (s0, s1), (r0, r1), (m0, m1))
            t["orientation"] = int(rng.integers(0, 2))
            t["bs_strand"] = int(rng.integers(0, 3))
            ts.append(t)
        # fixed trims up to beyond the middle of a read: the right trim's mirrored bases
        lt = tuple(int(v) for v in rng.integers(0, 90, 2)) if rng.random() < 0.4 else (0, 0)
        rt = tuple(int(v) for v in rng.integers(0, 90, 2)) if rng.random() < 0.4 else (0, 0)
        n_err += _both(caller, ts, left_trim=lt, right_trim=rt, min_qual=int(rng.integers(1, 44))) is None
    assert n_err < 125  # most lists are valid


def test_tiny_long_and_oddly_placed_reads(caller):
    """The copy kernel moves a plain read as dwords — lane l bytes 4l .. 4l + 3, the last lane the read's last four bytes over its
    neighbour's, 256 bytes a round — and sends what does not fit that (shorter than four bytes, a right trim, a mark more than 255
    bytes in) the long way: lengths around every one of those limits, every quality incl. 0 and 63, fixed trims short and long
    (beyond the read's middle, beyond the read), single ends, empty reads, overlapping mates of equal span (the mean-quality tie),
    output offsets of every alignment; with the read profile too."""
    from bs_call_amd.caller import ReadProfile

    rng = np.random.default_rng(777)
    lens = [1, 2, 3, 4, 5, 6, 7, 8, 9, 31, 63, 64, 65, 127, 128, 129, 252, 253, 254, 255, 256, 257, 258, 259, 260, 300, 511, 512, 513, 514, 700, 1023, 1025]
    for trial in range(24):
        ts, p0 = [], 400
        for i in range(int(rng.integers(20, 140))):
            n0, n1 = int(rng.choice(lens)), int(rng.choice(lens))
            qmode = rng.random()
            mk = lambda n: [int(rng.integers(0, 4)) | ((63 if qmode < 0.1 else (0 if qmode < 0.2 else int(rng.integers(0, 64)))) << 2) for _ in range(n)]
            p0 += int(rng.integers(0, 40))
            kind = rng.random()
            if kind < 0.2:
                t = T.tpl((p0, 0), (n0, 0), (mk(n0), None))
            elif kind < 0.3:
                t = T.tpl((0, p0), (0, n1), (None, mk(n1)))
            elif kind < 0.5:  # overlapping mates, often of equal span
                n1 = n0 if rng.random() < 0.6 else n1
                t = T.tpl((p0, p0 + int(rng.integers(0, n0 + 1))), (n0, n1), (mk(n0), mk(n1)))
            else:
                t = T.tpl((p0, p0 + n0 + int(rng.integers(1, 50))), (n0, n1), (mk(n0), mk(n1)))
            t["orientation"] = int(rng.integers(0, 2))
            t["bs_strand"] = int(rng.integers(0, 3))
            ts.append(t)
        lt = tuple(int(v) for v in rng.choice([0, 1, 3, 5, 100, 254, 255, 256, 257, 300, 2000], 2))
        rt = tuple(int(v) for v in rng.choice([0, 0, 0, 1, 4, 130, 600], 2))
        kw = dict(left_trim=lt, right_trim=rt, min_qual=int(rng.choice([0, 1, 20, 63, 64, 70])))
        got = _both(caller, ts, **kw)
        assert got is not None
        # the read profile over the same templates (reads at the block's first position in every third trial)
        raw, seq, ms = T.to_arrays(ts)
        h_tpl = got[0]
        pos = [int(p) for p in h_tpl["pos"].ravel() if p]
        x = max(1, min(pos) - (0 if trial % 3 == 0 else 2))
        y = int((h_tpl["pos"].astype(np.int64) + h_tpl["len"]).max()) + 3
        ref = rng.integers(0, 5, size=y - x + 3).astype(np.uint8)
        pf_h, pf_d = ReadProfile(cap=4096), ReadProfile(cap=4096)
        prepare_templates(raw, seq, ms, profile=pf_h, x=x, ref=ref, **kw)
        d_tpl, d_seq, d_st = caller.prepare_templates_device(raw, seq, ms, profile=pf_d, x=x, ref=ref, **kw)
        assert d_seq.tobytes() == got[1].tobytes() and d_tpl.tobytes() == got[0].tobytes() and d_st.tobytes() == got[2].tobytes()
        assert pf_d.used == pf_h.used and pf_d.counts[: pf_d.used].tobytes() == pf_h.counts[: pf_h.used].tobytes(), trial


def test_device_equals_python_restatement(caller):
    rng = np.random.default_rng(99)
    done = 0
    while done < 40:
        ts = []
        for _ in range(int(rng.integers(1, 10))):
            r0, m0, s0 = T._random_read(rng)
            r1, m1, s1 = T._random_read(rng)
            p0 = int(rng.integers(50, 5000))
            ts.append(T.tpl((p0, max(1, p0 + int(rng.integers(-s1 - 5, s0 + 30)))), (s0, s1), (r0, r1), (m0, m1), orientation=int(rng.integers(0, 2))))
        kw = dict(left_trim=(int(rng.integers(0, 5)), 0), right_trim=(0, int(rng.integers(0, 5))), min_qual=20)
        try:
            p, pst = py_prep.prepare(ts, **kw)
        except py_prep.PrepError:
            continue
        raw, seq, ms = T.to_arrays(ts)
        d_tpl, d_seq, d_st = caller.prepare_templates_device(raw, seq, ms, **kw)
        got = [{"pos": [int(v) for v in o["pos"]], "reads": [d_seq[int(o["off"][k]) : int(o["off"][k]) + int(o["len"][k])].tolist() for k in range(2)],
                "mapq": [int(v) for v in o["mapq"]], "orientation": int(o["orientation"]), "bs_strand": int(o["bs_strand"])} for o in d_tpl]
        assert got == p and {f: int(d_st[f]) for f in d_st.dtype.names} == pst
        done += 1


def test_block_of_generated_reads_through_the_device_prep_feeds_the_accumulate_stage(caller, oracle):
    """A 200 000-position block of generated pairs handed over RAW (no indels: the span is the length; trimmed stretches, empty reads, single ends): prepared on
    the device, the pile-up of the prepared reads — still on the device — equals the oracle's over the host-prepared ones."""
    import torch

    from bs_call_amd.abi import RAW_TEMPLATE

    tpl, seq = B.synth_reads_host(88172645463325252 + 5, 3000, 200_000, 30)
    raw = np.zeros(len(tpl), dtype=RAW_TEMPLATE)
    for f in ("pos", "len", "off", "mapq", "orientation", "bs_strand"):
        raw[f] = tpl[f]
    raw["reference_span"] = tpl["len"]
    ms = np.zeros(0, dtype=B.abi.MISMS)
    h_tpl, h_seq, h_st = prepare_templates(raw, seq, ms)
    d_tpl, d_seq, used, d_st = caller.prepare_templates_device(raw, seq, ms, keep_on_device=True)
    assert used == h_seq.size and d_st.tobytes() == h_st.tobytes() and int(h_st["base_trim"]) > 0
    assert d_seq.cpu().numpy()[:used].tobytes() == h_seq.tobytes()
    assert d_tpl.cpu().numpy()[: len(raw) * 40].tobytes() == h_tpl.tobytes()
    x = max(1, int(min(p for p in h_tpl["pos"].ravel() if p)) - 2)
    y = int((h_tpl["pos"].astype(np.int64) + h_tpl["len"]).max()) - 1
    rc, exp = oracle.accumulate(h_tpl, h_seq, x, y, 20)
    assert rc == 0
    n_pad = (y - x + 1 + 63) // 64 * 64
    d_cts = torch.zeros(n_pad * 104, dtype=torch.uint8, device=d_seq.device)
    caller.accumulate_device(d_tpl.data_ptr(), len(raw), d_seq.data_ptr(), used, x, y, d_cts.data_ptr(), None)
    caller.block_status(None)
    torch.cuda.synchronize()
    assert d_cts.cpu().numpy()[: (y - x + 1) * 104].tobytes() == exp.tobytes()


def test_block_records_from_raw_templates(caller):
    """bsc_block_records_raw: raw templates with indels, soft clips and overlapping mates in, packed records out — the records (and
    the statistics) of bsc_prepare_templates on the host followed by bsc_block_records."""
    rng = np.random.default_rng(4711)
    ts, p0 = [], 2000
    for i in range(3000):
        r0, m0, s0 = T._random_read(rng, 20)
        r1, m1, s1 = T._random_read(rng, 20)
        p0 += int(rng.integers(0, 6))
        kind = rng.random()
        if kind < 0.1:
            t = T.tpl((p0, 0), (s0, 0), (r0, None), (m0, ()))
        else:
            t = T.tpl((p0, p0 + int(rng.integers(0, s0 + 40))), (s0, s1), (r0, r1), (m0, m1))
        t["orientation"] = int(rng.integers(0, 2))
        t["bs_strand"] = int(rng.integers(1, 3))
        ts.append(t)
    ok = []
    for t in ts:  # keep the templates the reference would not abort on
        try:
            py_prep.prepare([t])
            ok.append(t)
        except py_prep.PrepError:
            pass
    raw, seq, ms = T.to_arrays(ok)
    kw = dict(left_trim=(2, 0), right_trim=(0, 3), min_qual=20)
    h_tpl, h_seq, h_st = prepare_templates(raw, seq, ms, **kw)
    x = max(1, int(min(p for p in raw["pos"].ravel() if p)) - 2)
    y = int(max((raw["pos"].astype(np.int64) + raw["reference_span"]).max(), (h_tpl["pos"].astype(np.int64) + h_tpl["len"]).max())) + 5
    ref = B.synth_ref_host(7, x, y - x + 3)
    caller.reset_site_stats()
    want = caller.block_records(h_tpl, h_seq, x, y, ref, with_stats=True).copy()
    st_want = caller.site_stats().copy()
    caller.reset_site_stats()
    got, st = caller.block_records_raw(raw, seq, ms, x, y, ref, with_stats=True, **kw)
    assert len(want) > 1000 and got.tobytes() == want.tobytes() and st.tobytes() == h_st.tobytes()
    assert caller.site_stats().tobytes() == st_want.tobytes()
    # ... and, with no product code in between, the records of the CPU oracle chain over the same raw templates
    o_tpl, o_seq, o_st = OC.prepare(ok, **kw)
    assert {f: int(st[f]) for f in st.dtype.names} == o_st
    OC.same_records(got, *OC.records(o_tpl, o_seq, x, y, ref))
    # a template the reference aborts on: refused with the host form's words, and the context goes on
    bad = list(ok[:50]) + [T.tpl((3000, 0), (30, 0), (T.read(30), None), (((3, 5, 4),), ()))] + list(ok[50:60])
    raw2, seq2, ms2 = T.to_arrays(bad)
    with pytest.raises(B.BscError, match="template 50 read 0: illegal soft clip"):
        caller.block_records_raw(raw2, seq2, ms2, x, y, ref, **kw)
    again, _ = caller.block_records_raw(raw, seq, ms, x, y, ref, **kw)
    assert again.tobytes() == want.tobytes()


def test_read_profile_on_the_device(caller):
    """The non-CpG read profile (meth_profile, src/meth_profile.c:48-77) made by the device form: the host form's counts and vector
    length, over several calls that grow the vector (incl. the clearing of what lies one past its old end), on random alignments with
    indels, clips and overlapping mates, reads at the block's first position (the walk that starts one code late) among them."""
    from bs_call_amd.caller import ReadProfile

    rng = np.random.default_rng(31337)
    pf_h, pf_d = ReadProfile(cap=4096), ReadProfile(cap=4096)
    for call in range(12):
        ts, p0 = [], 500
        for i in range(int(rng.integers(5, 400))):
            qlo = 20
            r0, m0, s0 = T._random_read(rng, qlo)
            r1, m1, s1 = T._random_read(rng, qlo)
            if call < 4:  # short reads first: later calls make the vector grow
                r0, m0, s0 = T.read(20 + call * 5, q=30), (), 20 + call * 5
                r1, m1, s1 = T.read(22 + call * 5, start=1, q=33), (), 22 + call * 5
            p0 += int(rng.integers(0, 9))
            kind = rng.random()
            if kind < 0.12:
                t = T.tpl((p0, 0), (s0, 0), (r0, None), (m0, ()))
            elif kind < 0.2:
                t = T.tpl((0, p0), (0, s1), (None, r1), ((), m1))
            else:
                t = T.tpl((p0, p0 + int(rng.integers(0, s0 + 30))), (s0, s1), (r0, r1), (m0, m1))
            t["orientation"] = int(rng.integers(0, 2))
            t["bs_strand"] = int(rng.integers(0, 3))
            try:
                py_prep.prepare([t])
                ts.append(t)
            except py_prep.PrepError:
                pass
        raw, seq, ms = T.to_arrays(ts)
        h_tpl, h_seq, _ = prepare_templates(raw, seq, ms)
        x = int(min(p for p in h_tpl["pos"].ravel() if p)) - (0 if call % 3 == 0 else 2)  # every third block starts AT its first read
        x = max(1, x)
        y = int((h_tpl["pos"].astype(np.int64) + h_tpl["len"]).max()) + 3
        ref = rng.integers(0, 5, size=y - x + 3).astype(np.uint8)
        kw = dict(left_trim=(int(rng.integers(0, 4)), 0), right_trim=(0, int(rng.integers(0, 4))))
        prepare_templates(raw, seq, ms, profile=pf_h, x=x, ref=ref, **kw)
        caller.prepare_templates_device(raw, seq, ms, profile=pf_d, x=x, ref=ref, **kw)
        # (what the host form leaves ONE PAST the vector's end — a reverse read's first base — is not part of the profile: the next
        # growth clears it, csrc/prep.c; the device form never writes it)
        assert pf_d.used == pf_h.used and pf_d.counts[: pf_d.used].tobytes() == pf_h.counts[: pf_h.used].tobytes(), call
        assert not pf_d.counts[pf_d.used :].any()
    assert pf_h.used > 60 and int(pf_h.counts[1 : pf_h.used].sum()) > 1000


@pytest.mark.parametrize("lengths", [(100, 100), (4, 5, 7, 100, 127), (128, 129, 130, 150), (100, 150, 255, 256), (100, 257, 300, 700), (3, 2, 1, 100)])
def test_read_profile_by_read_length(caller, lengths):
    """The profile pass keeps the counts of a read's positions in registers — two reads to a wave up to 128 positions, one up to 256, byte
    by byte beyond — and switches between the forms from one group of 64 reads to the next: plain reads (nothing cut or padded) of lengths
    around every one of those limits, both reads of a template, the three strands, random qualities (below 20 and 63 among them), fixed
    trims that move the positions, soft clips, overlapping mates, a block that starts AT its first read; several calls that grow the vector."""
    from bs_call_amd.caller import ReadProfile

    rng = np.random.default_rng(4242 + sum(lengths))
    pf_h, pf_d = ReadProfile(cap=2048), ReadProfile(cap=2048)
    n_counts = 0
    for call in range(5):
        ts, p0 = [], 300
        n_t = int(rng.integers(40, 700))
        for i in range(n_t):
            la, lb = int(rng.choice(lengths)), int(rng.choice(lengths))
            if call == 0:
                la, lb = min(la, 60), min(lb, 60)  # the vector grows in the later calls
            mk = lambda n: [T.b(int(v), int(q)) for v, q in zip(rng.integers(0, 4, n), rng.choice([5, 19, 20, 21, 30, 40, 62, 63, 0], n))]
            r0, r1 = mk(la), mk(lb)
            p0 += int(rng.integers(0, 7))
            m0 = m1 = ()
            if rng.random() < 0.1 and la > 12:  # a soft clip at the read's start: the positions shift
                c = int(rng.integers(1, 6))
                m0 = ([T.SOFT, 0, c],)
                sa = la - c
            else:
                sa = la
            kind = rng.random()
            if kind < 0.15:
                t = T.tpl((p0, 0), (sa, 0), (r0, None), (m0, ()))
            elif kind < 0.3:
                t = T.tpl((0, p0), (0, lb), (None, r1), ((), m1))
            else:
                t = T.tpl((p0, p0 + int(rng.integers(0, sa + 40))), (sa, lb), (r0, r1), (m0, m1))
            t["orientation"] = int(rng.integers(0, 2))
            t["bs_strand"] = int(rng.integers(0, 3)) if rng.random() < 0.3 else int(rng.integers(1, 3))
            try:
                py_prep.prepare([t])
                ts.append(t)
            except py_prep.PrepError:
                pass
        raw, seq, ms = T.to_arrays(ts)
        kw = dict(left_trim=(int(rng.integers(0, 4)), int(rng.integers(0, 3))), right_trim=(0, int(rng.integers(0, 3)) if call % 2 else 0))
        h_tpl, h_seq, _ = prepare_templates(raw, seq, ms, **kw)
        x = max(1, int(min(p for p in h_tpl["pos"].ravel() if p)) - (0 if call % 2 == 0 else 2))
        y = int((h_tpl["pos"].astype(np.int64) + h_tpl["len"]).max()) + 3
        ref = rng.integers(0, 5, size=y - x + 3).astype(np.uint8)
        ref[rng.random(len(ref)) < 0.5] = 2  # plenty of C ...
        ref[rng.random(len(ref)) < 0.3] = 3  # ... and G
        prepare_templates(raw, seq, ms, profile=pf_h, x=x, ref=ref, **kw)
        d_tpl, d_seq, d_st = caller.prepare_templates_device(raw, seq, ms, profile=pf_d, x=x, ref=ref, **kw)
        assert d_tpl.tobytes() == h_tpl.tobytes() and d_seq.tobytes() == h_seq.tobytes()
        assert pf_d.used == pf_h.used and pf_d.counts[: pf_d.used].tobytes() == pf_h.counts[: pf_h.used].tobytes(), (call, lengths)
        assert not pf_d.counts[pf_d.used :].any()
        n_counts = int(pf_h.counts[1 : pf_h.used].sum())
    assert n_counts > 2000 and pf_h.used >= min(max(lengths), 128)


def test_mates_of_equal_span_are_decided_by_their_mean_qualities(caller):
    """handle_overlap's tie-break (src/al_utils.c:191-203): overlapping mates of EQUAL span — the common pair of a short-insert library —
    are decided by the mean of the qualities that are not 63.  The device sums sixteen qualities a load, byte-parallel: read lengths
    around 4 and 16, qualities 63 and 0 anywhere, fixed trims that mark both ends, means that tie and means one apart."""
    rng = np.random.default_rng(6363)
    for trial in range(60):
        ts = []
        for _ in range(int(rng.integers(20, 80))):
            n = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 100, 150]))
            qs = []
            for k in range(2):
                mode = rng.random()
                if mode < 0.25:
                    q = np.full(n, int(rng.integers(0, 44)))
                elif mode < 0.5:
                    q = rng.integers(28, 32, n)  # means that tie or differ by one
                else:
                    q = rng.integers(0, 44, n)
                q = np.where(rng.random(n) < 0.1, 63, q)
                q = np.where(rng.random(n) < 0.05, 0, q)
                qs.append(q)
            r0 = [T.b(int(rng.integers(0, 4)), int(v)) for v in qs[0]]
            r1 = [T.b(int(rng.integers(0, 4)), int(v)) for v in qs[1]]
            p0 = int(rng.integers(100, 4000))
            p1 = p0 + int(rng.integers(-(n - 1), n))  # overlapping, either read first
            if p1 < 1:
                p1 = p0
            t = T.tpl((p0, p1), (n, n), (r0, r1))
            t["orientation"] = int(rng.integers(0, 2))
            ts.append(t)
        lt = (int(rng.integers(0, 6)), int(rng.integers(0, 6)))
        rt = (int(rng.integers(0, 6)), int(rng.integers(0, 6)))
        _both(caller, ts, left_trim=lt, right_trim=rt, min_qual=int(rng.choice([0, 13, 20])))
