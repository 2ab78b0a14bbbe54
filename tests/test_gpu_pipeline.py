"""BAM -> BCF end to end on the GPU (bs_call_amd.pipeline: reader -> read pre-processing -> bsc_block_records -> BCF
encoder) against the CPU oracle chain over the same file: oracle/py_bam.py (reader) -> oracle/py_prep.py (pre-processing) ->
the C oracle's accumulate / call / record formation -> oracle/py_bcf.py (encoder).  The BCF streams must be identical."""
import importlib.util
import json
import os
import struct

import numpy as np
import pytest

import bs_call_amd as B
from bs_call_amd import pipeline, vcf
from oracle import py_bam, py_bcf, py_prep

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("make_bam", os.path.join(ROOT, "tools", "make_bam.py"))
W = importlib.util.module_from_spec(spec)
spec.loader.exec_module(W)


def _gc_bins(codes):
    """load_sequence's bins (src/read_reference.c:66-104): from the first A/C/G/T base, 100 at a time, 255 with an N."""
    k = int(np.argmax((codes >= 1) & (codes <= 4)))
    bins = []
    for b0 in range(k, len(codes) - 99, 100):
        w = codes[b0 : b0 + 100]
        bins.append(int(((w == 2) | (w == 3)).sum()) if ((w >= 1) & (w <= 4)).all() else 255)
    return k + 1, bins


def _oracle_records(oracle, tables, libm_exact, bam, reference, gc=None, min_qual=20):
    """The written records of every block, as dicts for py_bcf.encode_record, through the CPU oracle chain; gc: a
    (4096, 101) array that receives the GC-by-coverage census of the positions that reached the printer."""
    text, refs, recs = py_bam.parse_bam(bam)
    out = []
    bins_of = {name: _gc_bins(codes) for name, codes in reference.items()}
    for tid, y, als in py_bam.read_input(recs):
        py_bam.check_block(als, y)
        name = refs[tid][0]
        x = als[0]["pos"][0] or als[0]["pos"][1]
        x = x - 2 if x > 2 else 1
        codes = reference[name]
        # get_sequence_string (src/get_sequence.c:35-48): positions at or beyond the contig's last one read as N
        ref = np.array([codes[p - 1] if p < len(codes) else 0 for p in range(x, y + 3)], dtype=np.uint8)
        prepared, _ = py_prep.prepare(als, min_qual=min_qual)
        tpl = np.zeros(len(prepared), dtype=B.TEMPLATE)
        seq = []
        for i, t in enumerate(prepared):
            tpl["pos"][i] = t["pos"]
            tpl["mapq"][i] = t["mapq"]
            tpl["orientation"][i] = t["orientation"]
            tpl["bs_strand"][i] = t["bs_strand"]
            for k in range(2):
                tpl["off"][i, k] = len(seq)
                tpl["len"][i, k] = len(t["reads"][k])
                seq += t["reads"][k]
        seq = np.array(seq, dtype=np.uint8)
        rc, pile = oracle.accumulate(tpl, seq, x, y, min_qual)
        assert rc == 0
        gtm, skip = oracle.call_sites(pile, ref[: y - x + 1], tables, oracle.LIBM if libm_exact else oracle.BSM, 1)
        core = oracle.vcf_block(gtm, skip, ref, x, reg_stop=len(codes))
        if gc is not None:
            start, bins = bins_of[name]
            for c, g in zip(core, gtm):
                pos = int(c["pos"])
                if pos >= start and (pos - start) // 100 < len(bins) and bins[(pos - start) // 100] <= 100:
                    gc[min(int(g["counts"].sum()), 4095), bins[(pos - start) // 100]] += 1
        for c, g in zip(core, gtm):
            if not c["emit"]:
                continue
            out.append((tid, dict(pos=int(c["pos"]), gt=int(c["gt"]), flt=int(c["flt"]), phred=int(c["phred"]), alt=bytes(c["alt"]).rstrip(b"\0"),
                                  ref=bytes(c["cx_ref"])[2:3], cx_ref=bytes(c["cx_ref"]), cx_gt=bytes(c["cx_gt"]), cg=bytes(c["cg"]),
                                  gt_enc=int(c["gt_enc"]), dp=int(c["dp"]), mq=int(g["mq"]), qd=int(c["qd"]), fs=int(c["fs"]),
                                  gl=[float(v) for v in c["gl"][: int(c["n_gl"])]], counts=[int(v) for v in g["counts"]], qual=[int(v) for v in g["qual"]])))
    return refs, out


def test_bam_to_bcf_equals_the_oracle_chain(tmp_path, oracle, tables, libm_exact):
    rng = np.random.default_rng(31)
    reference = {"chrA": rng.integers(1, 5, 30_000).astype(np.uint8), "chrB": rng.integers(1, 5, 12_000).astype(np.uint8)}
    reference["chrA"][5_000:5_400] = 0  # an N run
    refs = [(k, len(v)) for k, v in reference.items()]
    recs = W.wgbs_records(rng, reference["chrA"], 0, 1500, het_every=500) + W.wgbs_records(rng, reference["chrB"], 1, 500, strand_tag="XG")
    # the XG tag is a Z tag for Bowtie / Bismark: rewrite the second contig's tags accordingly
    for r in recs:
        if r["tid"] == 1:
            r["aux"] = W.aux_str("XG", "CT" if r["aux"][3:4] == b"C" else "GA")
    bam, bcf, rep = str(tmp_path / "in.bam"), str(tmp_path / "out.bcf"), str(tmp_path / "report.json")
    W.write_bam(bam, refs, recs)
    res = pipeline.run(bam, reference, bcf, sample="S1", report_path=rep, date=(3, 10, 2026), compressed=False)
    # the read pre-processing ran on the device (bsc_block_records_raw); on the host (round 4's split) the same bytes come out
    bcf_h, rep_h = str(tmp_path / "host_prep.bcf"), str(tmp_path / "host_prep.json")
    pipeline.run(bam, reference, bcf_h, sample="S1", report_path=rep_h, date=(3, 10, 2026), compressed=False, host_prep=True)
    assert open(bcf_h, "rb").read() == open(bcf, "rb").read() and open(rep_h).read() == open(rep).read()
    # ... and the BCF encoding too (bsc_block_bcf_raw); with the packed records encoded on the host, the same bytes
    bcf_e, rep_e = str(tmp_path / "host_bcf.bcf"), str(tmp_path / "host_bcf.json")
    pipeline.run(bam, reference, bcf_e, sample="S1", report_path=rep_e, date=(3, 10, 2026), compressed=False, host_bcf=True)
    assert open(bcf_e, "rb").read() == open(bcf, "rb").read() and open(rep_e).read() == open(rep).read()
    assert res["blocks"] >= 2 and res["records"] > 5_000 and res["contigs"] == ["chrA", "chrB"]
    # ---- the oracle chain, encoded by the independent Python encoder ----
    gc = np.zeros((4096, 101), dtype=np.uint64)
    orefs, orecs = _oracle_records(oracle, tables, libm_exact, bam, reference, gc)
    assert orefs == refs and len(orecs) == res["records"]
    stream = open(bcf, "rb").read()
    hdr = vcf.header_text(refs, "S1", date=(3, 10, 2026)).encode() + b"\0"
    assert stream[:9] == b"BCF\x02\x02" + struct.pack("<I", len(hdr)) and stream[9 : 9 + len(hdr)] == hdr
    exp = b"".join(py_bcf.encode_record(d, tid) for tid, d in orecs)
    assert stream[9 + len(hdr) :] == exp
    # ---- the report ----
    text = open(rep).read()
    d = json.loads(text.replace('": \\n\\t\\t\\t}', '": {\\n\\t\\t\\t}'))
    rl = d["filterStats"]["ReadLevel"]
    # every read is accounted for: passed, dropped as the duplicate of a template with the same positions and strand, or the
    # mate of such a template arriving to find nobody waiting
    assert rl["Passed"]["Reads"] + rl.get("Duplicate", {"Reads": 0})["Reads"] + rl.get("PairNotFound", {"Reads": 0})["Reads"] == 2 * 2000
    assert rl["Passed"]["Reads"] > 3800 and d["totalStats"]["SNPS"]["All"] == res["records"]
    prof = d["totalStats"]["methylation"]["NonCpGreadProfile"]
    assert len(prof) == 100 and sum(map(sum, prof)) > 10_000  # read positions 0 .. 99
    # flt_tab (src/init_param.c:57-70) puts the converting strand's observations (C / T on C2T reads, G / A on G2A reads) in
    # counts 0 / 1 and the other strand's in 2 / 3: the generator converts non-CpG cytosines with p = 120 / 128 and miscalls 0.5 %
    conv = sum(p[1] for p in prof) / max(1, sum(p[0] + p[1] for p in prof))
    miss = sum(p[3] for p in prof) / max(1, sum(p[2] + p[3] for p in prof))
    assert 0.90 < conv < 0.97 and miss < 0.01
    assert sum(v["SNPS"]["All"] for v in d["contigStats"].values()) == res["records"]
    got_gc = {int(k): v for k, v in d["totalStats"]["coverage"]["GC"].items()}
    assert int(gc.sum()) > 30_000 and all(got_gc[cv] == [int(v) for v in gc[cv]] for cv in got_gc)
    assert sum(sum(v) for v in got_gc.values()) == int(gc.sum())


def test_bam_to_bcf_with_other_model_parameters(tmp_path, oracle, libm_exact):
    """run(..., under_conv=, over_conv=, min_qual=) builds the caller from THOSE values (src/init_param.c:26-31,
    src/parse_args.c:126-136): records = the oracle chain with the same parameters, not with the defaults, and the header
    names them; a supplied caller's own parameters win and a contradicting explicit value is refused."""
    rng = np.random.default_rng(77)
    reference = {"chrA": rng.integers(1, 5, 20_000).astype(np.uint8)}
    refs = [("chrA", 20_000)]
    recs = W.wgbs_records(rng, reference["chrA"], 0, 1200, het_every=300)
    bam, bcf = str(tmp_path / "in.bam"), str(tmp_path / "out.bcf")
    W.write_bam(bam, refs, recs)
    uc, oc, mq = 0.03, 0.11, 27
    res = pipeline.run(bam, reference, bcf, sample="S2", date=(4, 10, 2026), compressed=False, under_conv=uc, over_conv=oc, min_qual=mq)
    tb = oracle.Tables(uc, oc, 2.0, mq)
    _, orecs = _oracle_records(oracle, tb, libm_exact, bam, reference, min_qual=mq)
    _, drecs = _oracle_records(oracle, oracle.Tables(), libm_exact, bam, reference)
    assert len(orecs) == res["records"]
    stream = open(bcf, "rb").read()
    hdr = vcf.header_text(refs, "S2", date=(4, 10, 2026), under_conv=uc, over_conv=oc, min_qual=mq).encode() + b"\0"
    assert stream[9 : 9 + len(hdr)] == hdr
    exp = b"".join(py_bcf.encode_record(d, tid) for tid, d in orecs)
    assert stream[9 + len(hdr) :] == exp
    assert exp != b"".join(py_bcf.encode_record(d, tid) for tid, d in drecs)  # the parameters matter on this input
    # a supplied caller: its parameters go into the header; a contradicting explicit value is an error
    with B.SiteCaller(under_conv=uc, over_conv=oc, min_qual=mq) as c:
        bcf2 = str(tmp_path / "out2.bcf")
        pipeline.run(bam, reference, bcf2, sample="S2", date=(4, 10, 2026), compressed=False, caller=c)
        assert open(bcf2, "rb").read() == stream
        with pytest.raises(ValueError):
            pipeline.run(bam, reference, bcf2, compressed=False, caller=c, under_conv=0.01)


@pytest.mark.parametrize("seed", [3, 5, 7, 4])  # 3, 5, 7 run to the end; 4 holds a pair the reference asserts on
def test_bam_to_bcf_on_random_alignments(tmp_path, oracle, tables, libm_exact, seed):
    """The same equality on files full of what the generator above never makes: overlapping mates, soft clips, insertions and
    deletions, single reads of either strand, duplicates, filtered records, Ns, two contigs."""
    from tests import test_bam as TB

    rng = np.random.default_rng(seed)
    reference = {name: rng.integers(1, 5, ln).astype(np.uint8) for name, ln in TB.REFS}
    recs = TB._random_records(rng, 700)
    bam, bcf = str(tmp_path / "in.bam"), str(tmp_path / "out.bcf")
    W.write_bam(bam, TB.REFS, recs)
    try:
        orefs, orecs = _oracle_records(oracle, tables, libm_exact, bam, reference)
    except (AssertionError, py_prep.PrepError):
        # input the reference asserts on (or walks out of bounds on): the library reports it
        with pytest.raises(B.BscError):
            pipeline.run(bam, reference, bcf, compressed=False, date=(3, 10, 2026))
        return
    res = pipeline.run(bam, reference, bcf, compressed=False, date=(3, 10, 2026))
    assert len(orecs) == res["records"] > 1000
    stream = open(bcf, "rb").read()
    l_text = struct.unpack_from("<I", stream, 5)[0]
    assert stream[9 + l_text :] == b"".join(py_bcf.encode_record(d, tid) for tid, d in orecs)


def test_plain_c_bam2bcf_equals_the_python_pipeline(tmp_path):
    """integration/bam2bcf.c — a gcc-built C program that walks BAM + FASTA -> BCF + report with the C ABI alone — writes the
    bytes bs_call_amd.pipeline.run writes (header in --benchmark-mode, fixed report date)."""
    import subprocess

    exe = os.path.join(ROOT, "bs_call_amd", "lib", "bam2bcf")
    assert os.path.exists(exe), "run `make demo`"
    rng = np.random.default_rng(77)
    reference = {"chrA": rng.integers(1, 5, 20_000).astype(np.uint8), "chrB": rng.integers(1, 5, 9_000).astype(np.uint8)}
    reference["chrA"][:37] = 0
    reference["chrB"][4_000:4_250] = 0
    refs = [(k, len(v)) for k, v in reference.items()]
    recs = W.wgbs_records(rng, reference["chrA"], 0, 900, het_every=400) + W.wgbs_records(rng, reference["chrB"], 1, 400)
    bam, fa = str(tmp_path / "in.bam"), str(tmp_path / "ref.fa")
    W.write_bam(bam, refs, recs)
    with open(fa, "w") as f:
        for name, codes in reference.items():
            f.write(">%s some description\n" % name)
            s = "".join("NACGT"[c] for c in codes)
            for o in range(0, len(s), 60):
                f.write(s[o : o + 60] + "\n")
    out_c, rep_c = str(tmp_path / "c.bcf"), str(tmp_path / "c.json")
    r = subprocess.run([exe, bam, fa, out_c, rep_c, "S9"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    out_p, rep_p = str(tmp_path / "p.bcf"), str(tmp_path / "p.json")
    # the C program reads the FASTA itself; hand the Python pipeline the same codes through the library's FASTA reader
    from bs_call_amd.bam import fasta_contig

    ref2 = {name: fasta_contig(fa, name) for name in reference}
    assert all((ref2[k] == reference[k]).all() for k in reference)
    res = pipeline.run(bam, ref2, out_p, sample="S9", report_path=rep_p, date=(1, 1, 2000), compressed=False, benchmark_mode=True)
    assert open(out_c, "rb").read() == open(out_p, "rb").read() and res["records"] > 3_000
    # the C program with the encoder on its own thread (BAM2BCF_HOST_BCF, round 5's first form) writes the same files
    out_h, rep_h = str(tmp_path / "h.bcf"), str(tmp_path / "h.json")
    r = subprocess.run([exe, bam, fa, out_h, rep_h, "S9"], capture_output=True, text=True, timeout=300, env=dict(os.environ, BAM2BCF_HOST_BCF="1"))
    assert r.returncode == 0, r.stderr + r.stdout
    assert open(out_h, "rb").read() == open(out_c, "rb").read() and open(rep_h).read() == open(rep_c).read()
    assert open(rep_c).read() == open(rep_p).read()
    assert r.stdout.strip() == "%d blocks, %d records written" % (res["blocks"], res["records"])
