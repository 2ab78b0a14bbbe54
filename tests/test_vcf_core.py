"""VCF record formation (reference src/print_vcf.c:32-381,529-594): the oracle's rule-generated tables against the
reference's literal tables (CPU), and the device kernel against the oracle's sequential sliding-window restatement
(GPU), byte for byte, including block edges, N bases in the context, skipped positions and region clipping."""
import json
import os

import numpy as np
import pytest

import bs_call_amd as B

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 88172645463325252


def test_oracle_tables_match_reference_literals(oracle):
    gold = json.load(open(os.path.join(HERE, "golden", "print_vcf_tables.json")))
    ref_alt, all_idx, gt_int, gt_flag = oracle.vcf_tables()
    for g in range(10):
        for r in range(5):
            assert b"".join(ref_alt[g, r]).rstrip(b"\0").decode() == gold["ref_alt"][g][r], (g, r)
            assert all_idx[g, r].tolist() == gold["all_idx"][g][r], (g, r)
            assert int(gt_int[g, r]) == gold["gt_int"][g][r], (g, r)
    assert sorted(map(list, np.argwhere(gt_flag == 1).tolist())) == gold["gt_flag_ones"]
    # contains-C / contains-G / IUPAC rules
    names = B.GENOTYPES
    assert [int("C" in n) for n in names] == gold["cflag"] and [int("G" in n) for n in names] == gold["gflag"]
    iupac = {"AA": "A", "AC": "M", "AG": "R", "AT": "W", "CC": "C", "CG": "S", "CT": "Y", "GG": "G", "GT": "K", "TT": "T"}
    assert "N" + "".join(iupac[n] for n in names) == gold["iupac"]


def _called_block(oracle, tables, seed, x, n, cov, flags=0):
    pile, ref = B.synth_pileup_host(seed, x, n + 2, cov, flags)
    out, skip = oracle.call_sites(pile[:n], ref[:n], tables, oracle.LIBM, -8)
    return out, skip, ref


def test_oracle_block_sanity(oracle, tables):
    out, skip, ref = _called_block(oracle, tables, SEED, 1000, 5000, 30)
    rec = oracle.vcf_block(out, skip, ref, 1000, all_positions=True)
    cov = skip == 0
    assert (rec["pos"][cov] == 1000 + np.flatnonzero(cov)).all() and (rec["pos"][~cov] == 0).all()
    assert (rec["emit"][cov] == 1).all()
    assert (rec["gt"][cov] == out["max_gt"][cov]).all()
    mid = slice(10, -10)
    assert (rec["ref_code"][mid][cov[mid]] == ref[:5000][mid][cov[mid]]).all()
    # default mode drops hom-ref AA on A and TT on T
    rec2 = oracle.vcf_block(out, skip, ref, 1000)
    drop = cov & (((out["max_gt"] == 0) & (ref[:5000] == 1)) | ((out["max_gt"] == 9) & (ref[:5000] == 4)))
    assert (rec2["emit"][drop] == 0).all() and (rec2["emit"][cov & ~drop] == 1).all()
    assert set(np.unique(rec["cg"])) <= {b"C", b"H", b"N", b"?", b".", b""}
    assert (rec["phred"][cov] <= 255).all() and (rec["n_gl"][cov & (rec["emit"] == 1)] >= 1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 63, 64, 65, 1000])
def test_small_blocks_and_edges(oracle, tables, n):
    with B.SiteCaller() as c:
        for x in (7, 1000):
            out, skip, ref = _called_block(oracle, tables, SEED + n, x, n, 30)
            for ap in (False, True):
                got = c.vcf_records(out, skip, ref, x, all_positions=ap)
                exp = oracle.vcf_block(out, skip, ref, x, all_positions=ap)
                assert got.tobytes() == exp.tobytes(), (n, x, ap)


@pytest.mark.gpu
def test_block_parity_with_N_skips_region_dbsnp(oracle, tables, libm_exact):
    rng = np.random.default_rng(3)
    with B.SiteCaller() as c:
        for cov, n, x in ((30, 200_000, 5_000), (10, 50_000, 123_456), (300, 10_000, 99)):
            out, skip, ref = _called_block(oracle, tables, SEED + cov, x, n, cov)
            ref = ref.copy()
            ref[rng.integers(0, n + 2, size=n // 50)] = 0  # scattered N bases: exercises the strncpy truncation
            skip = skip.copy()
            skip[rng.integers(0, n, size=n // 20)] = 1  # scattered uncovered positions
            db = rng.choice([0, 1, 3], size=n, p=[0.9, 0.05, 0.05]).astype(np.uint8)
            for kw in (dict(), dict(all_positions=True), dict(reg_start=x + 1000, reg_stop=x + n // 2), dict(dbsnp=db)):
                got = c.vcf_records(out, skip, ref, x, **kw)
                exp = oracle.vcf_block(out, skip, ref, x, **kw)
                if not libm_exact:  # phred goes through exp/log: allow the libm difference there only
                    assert (np.abs(got["phred"].astype(int) - exp["phred"].astype(int)) <= 1).all()
                    got["phred"] = exp["phred"]
                    got["qd"] = exp["qd"]
                    got["flt"] = exp["flt"]
                bad = np.flatnonzero(got.view(np.uint8).reshape(n, 64).tobytes() != exp.tobytes()) if False else None
                if got.tobytes() != exp.tobytes():
                    d = np.flatnonzero((got.view(np.uint8).reshape(n, 64) != exp.view(np.uint8).reshape(n, 64)).any(axis=1))
                    raise AssertionError("records differ at %s: got %r exp %r" % (d[:5], got[d[0]], exp[d[0]]))
            # strided gt_vcf input gives the same records
            raw = np.zeros((n, 208), dtype=np.uint8)
            raw[:, :200] = out.view(np.uint8).reshape(n, 200)
            got2 = c.vcf_records(raw, skip, ref, x)
            assert got2.tobytes() == oracle.vcf_block(out, skip, ref, x).tobytes()


@pytest.mark.gpu
def test_device_chain_call_then_vcf(oracle, tables, libm_exact):
    """HBM-resident chain: pile-up -> gt_meth -> VCF records without touching the host in between."""
    import torch

    n, x, cov = 300_000, 10_000, 30
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
        torch.cuda.synchronize()
        got = d_vcf.cpu().numpy().view(B.VCF_CORE)
    pile, ref = B.synth_pileup_host(SEED + 9, x, n + 2, cov)
    out, skip = oracle.call_sites(pile[:n], ref[:n], tables, oracle.LIBM if libm_exact else oracle.BSM, -8)
    exp = oracle.vcf_block(out, skip, ref, x)
    if libm_exact:
        assert got.tobytes() == exp.tobytes()
    else:
        assert (got["gt"] == exp["gt"]).all() and (got["emit"] == exp["emit"]).all()


# ---- an independently written renderer of the same line, the checker of the library's C formatter ---------------------
def _gl(v):
    return "%g" % float(v)


def _py_format_record(core, gtm, contig, rs_id="."):
    from bs_call_amd.abi import GENOTYPES
    from bs_call_amd.vcf import CS_STR, FLT_NAMES

    gt = int(core["gt"])
    het = GENOTYPES[gt][0] != GENOTYPES[gt][1]
    flt = int(core["flt"])
    alt = core["alt"].decode()
    cols = [contig, str(int(core["pos"])), rs_id, core["cx_ref"].decode()[2], ",".join(alt) if alt else ".",
            str(int(core["phred"])), "PASS" if flt == 0 else ("mac1" if flt & 128 else "fail"),
            "CX=" + core["cx_ref"].decode()]
    enc = int(core["gt_enc"])
    a, b = (enc >> 4 >> 1) - 1, ((enc & 15) >> 1) - 1
    # the reference leaves each name's NUL in the FT buffer (src/print_vcf.c:289-293): text shows the first failed filter only
    ft = next(n for i, n in enumerate(FLT_NAMES) if flt >> i & 1) if flt & 15 else "PASS"
    counts = [int(c) for c in gtm["counts"]]
    amq = [str(int(q)) for c, q in zip(counts, gtm["qual"]) if c > 0]
    keys = ["GT", "FT", "DP", "MQ", "GQ", "QD", "GL", "MC8"]
    vals = ["%d/%d" % (a, b), ft, str(int(core["dp"])), str(int(gtm["mq"])), str(int(core["phred"])), str(int(core["qd"])),
            ",".join(_gl(v) for v in core["gl"][: int(core["n_gl"])]), ",".join(map(str, counts))]
    if amq:
        keys.append("AMQ")
        vals.append(",".join(amq))
    keys += ["CS", "CG", "CX"]
    vals += [CS_STR[gt], core["cg"].decode(), core["cx_gt"].decode()]
    if het:
        keys.append("FS")
        vals.append(str(int(core["fs"])))
    return "\t".join(cols + [":".join(keys), ":".join(vals)])


def _py_format_block(cores, gtms, contig):
    return [_py_format_record(c, g, contig) for c, g in zip(cores, gtms) if c["emit"]]


def test_text_rendering_of_oracle_records(oracle, tables):
    """The host formatter on records made by the CPU oracle (no GPU needed): field layout and integer content."""
    from bs_call_amd import vcf

    assert vcf.CS_STR == tuple(json.load(open(os.path.join(HERE, "golden", "print_vcf_tables.json")))["cs_str"])
    out, skip, ref = _called_block(oracle, tables, SEED, 1000, 4000, 30)
    rec = oracle.vcf_block(out, skip, ref, 1000)
    lines = vcf.format_block(rec, out, "chrS")
    assert len(lines) == int(rec["emit"].sum()) > 1000
    for ln in lines[:200]:
        f = ln.split("\t")
        assert len(f) == 10 and f[0] == "chrS" and f[6] in ("PASS", "fail", "mac1") and f[7].startswith("CX=")
        keys, vals = f[8].split(":"), f[9].split(":")
        assert len(keys) == len(vals) and keys[:8] == ["GT", "FT", "DP", "MQ", "GQ", "QD", "GL", "MC8"]
        kv = dict(zip(keys, vals))
        i = int(f[1]) - 1000
        assert int(kv["GQ"]) == int(f[5]) == int(rec["phred"][i]) and len(kv["MC8"].split(",")) == 8
        assert sum(map(int, kv["MC8"].split(",")[:4])) == int(kv["DP"])
        assert ("FS" in kv) == bool(B.GT_HET[rec["gt"][i]])


def test_c_formatter_equals_python_formatter(oracle, tables):
    """bsc_vcf_format (host C, the library's one formatter) renders the lines the independent Python renderer above does."""
    from bs_call_amd import vcf

    out, skip, ref = _called_block(oracle, tables, SEED + 4, 5000, 20_000, 30)
    rec = oracle.vcf_block(out, skip, ref, 5000, all_positions=True)
    a = _py_format_block(rec, out, "chr7")
    b = vcf.format_block(rec, out, "chr7")
    assert len(a) == len(b) == int(rec["emit"].sum())
    assert a == b


def test_packed_records_render_the_same_lines(oracle, tables):
    """bsc_vcf_format_rec on a packed record (core + the gt_meth fields the encoder reads) == bsc_vcf_format on the pair."""
    from bs_call_amd import vcf

    out, skip, ref = _called_block(oracle, tables, SEED + 44, 1000, 3000, 30)
    core = oracle.vcf_block(out, skip, ref, 1000)
    sel = core["emit"] == 1
    rec = np.zeros(int(sel.sum()), dtype=B.VCF_REC)
    rec["core"], rec["counts"], rec["qual"] = core[sel], out["counts"][sel], out["qual"][sel]
    rec["mq"], rec["aq"], rec["max_gt"] = out["mq"][sel], out["aq"][sel], out["max_gt"][sel]
    a, b = vcf.format_records_c(rec, "chr1"), vcf.format_block_c(core, out, "chr1")
    assert len(a) == int(sel.sum()) > 100 and a == b
