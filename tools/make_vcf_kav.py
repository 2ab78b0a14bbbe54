#!/usr/bin/env python3
"""Writes tests/golden/vcf_kav.json: hand-worked records of the reference's VCF record formation.

Every EXPECTED value below is a literal worked out by hand from the cited line of the reference (src/print_vcf.c) — not
computed by any implementation in this repository.  This script only spares typing the inputs: it expands a compact
description of a small block (called genotypes, counts, the log10 posterior of the call, mapping quality, Fisher
log10 p, reference bases) into full gt_meth records.  tests/test_vcf_kav.py runs the fixture through oracle/py_vcf.py,
oracle/orc_vcf.c and (on the GPU box) the record kernels.

How a called genotype is planted: gt_prob[g] = the stated log10 posterior (<= 0); every other genotype gets
-30 - index (distinct, far below), unless a case states its gt_prob vector explicitly (the GL cases).
phred by hand (:140-148): z1 = 10^gt_prob[g]; z1 >= 1 -> 255, else (int)(-10 log10(1 - z1)).  Used values:
  gt_prob =  0        -> z1 = 1                              -> 255
  gt_prob = -0.012416 -> z1 = 0.971816, 1 - z1 = 0.028184    -> -10 log10 = 15.50 -> 15
  gt_prob = -0.5      -> z1 = 0.316228, 1 - z1 = 0.683772    -> 1.65 -> 1
  gt_prob = -0.0004345-> z1 = 0.9990,   1 - z1 = 0.0010      -> 30.0x: avoided (boundary); not used
  gt_prob = -0.001    -> z1 = 0.997700, 1 - z1 = 0.0023      -> 26.38 -> 26
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = {"AA": 0, "AC": 1, "AG": 2, "AT": 3, "CC": 4, "CG": 5, "CT": 6, "GG": 7, "GT": 8, "TT": 9}
BASE = {"N": 0, "A": 1, "C": 2, "G": 3, "T": 4}


def site(gt, lp=0.0, counts=(0, 0, 0, 0, 0, 0, 0, 0), mq=60, fisher=0.0, gp=None, quals=None):
    """One covered position calling genotype `gt` (name) with log10 posterior lp; gp overrides gt_prob completely."""
    if gp is None:
        gp = [-30.0 - i for i in range(10)]
        gp[G[gt]] = lp
    q = quals or [30 if c else 0 for c in counts]
    return {"skip": 0, "counts": list(counts), "qual": q, "gt_prob": list(gp), "fisher_strand": fisher, "mq": mq, "aq": 30,
            "max_gt": G[gt] if gt else 0}


NONE = {"skip": 1, "counts": [0] * 8, "qual": [0] * 8, "gt_prob": [0.0] * 10, "fisher_strand": 0.0, "mq": 0, "aq": 0, "max_gt": 0}
A30 = (30, 0, 0, 0, 0, 0, 0, 0)  # 30 non-informative A reads: DP = 30
C30 = (0, 30, 0, 0, 0, 0, 0, 0)
G30 = (0, 0, 30, 0, 0, 0, 0, 0)
T30 = (0, 0, 0, 30, 0, 0, 0, 0)
cases = []


def case(name, line, ref, sites, expect, x=100, dbsnp=None, all_positions=False, reg=(1, 0xFFFFFFFF)):
    """ref: reference bases of x .. x + n + 1 as letters; expect: {position: {field: value}} (only stated fields are checked);
    a position mapped to None must have NO record."""
    assert len(ref) == len(sites) + 2, name
    cases.append({"name": name, "decided_by": "src/print_vcf.c:" + line, "x": x, "ref": [BASE[c] for c in ref], "sites": sites,
                  "dbsnp": dbsnp, "all_positions": all_positions, "reg_start": reg[0], "reg_stop": reg[1],
                  "expect": {str(k): v for k, v in expect.items()}})


def flank(center_sites, left=("TT", "TT"), right=("TT", "TT")):
    """centre sites between two TT calls either side (on reference T those are never written, :139, but they ARE called)."""
    return [site(g, 0.0, T30) for g in left] + center_sites + [site(g, 0.0, T30) for g in right]


# ---- FILTER bits (:185-217) ------------------------------------------------------------------------------------------
# position 102 is the centre; reference TTCTTTT...: ref C at the centre, calls are CT (het) or CC
case("PASS", "185-190", "TTCTTTT", flank([site("CC", 0.0, C30)]),
     {102: {"emit": 1, "flt": 0, "phred": 255, "qd": 8, "dp": 30, "fs": 0, "gt": 4, "ref_code": 2, "alt": "", "gt_enc": 0x22, "n_gl": 1}})
case("q20 alone", "186", "TTCTTTT", flank([site("CC", -0.012416, (0, 5, 0, 0, 0, 0, 0, 0))]),
     {102: {"emit": 1, "flt": 1, "phred": 15, "qd": 3, "dp": 5}})  # 15 / 5 = 3 >= 2: only q20
case("qd2 alone", "187", "TTCTTTT", flank([site("CC", 0.0, (0, 200, 0, 0, 0, 0, 0, 0))]),
     {102: {"emit": 1, "flt": 2, "phred": 255, "qd": 1, "dp": 200}})  # 255 / 200 = 1 < 2
case("fs60 alone", "188", "TTCTTTT", flank([site("CT", 0.0, (0, 15, 0, 15, 0, 0, 0, 0), fisher=-6.2)]),
     {102: {"emit": 1, "flt": 4, "fs": 62, "gt": 6, "alt": "T", "gt_enc": 0x24}})  # (int)(62 + 0.5) = 62 > 60
case("fs exactly 60 passes", "188", "TTCTTTT", flank([site("CT", 0.0, (0, 15, 0, 15, 0, 0, 0, 0), fisher=-5.96)]),
     {102: {"emit": 1, "flt": 0, "fs": 60}})  # (int)(59.6 + 0.5) = 60, not > 60; both alleles >= 2: no mac1
case("mq40 alone", "189", "TTCTTTT", flank([site("CC", 0.0, C30, mq=39)]), {102: {"emit": 1, "flt": 8}})
case("q20 + qd2 + mq40", "186-189", "TTCTTTT", flank([site("CC", -0.5, C30, mq=10)]),
     {102: {"emit": 1, "flt": 11, "phred": 1, "qd": 0}})  # 1 / 30 = 0
case("a failed record is not tested for mac1", "190-191", "TTCTTTT", flank([site("CT", 0.0, (0, 15, 0, 1, 0, 0, 0, 0), mq=39)]),
     {102: {"emit": 1, "flt": 8}})
# ---- mac1 per genotype (:191-214) ------------------------------------------------------------------------------------
case("mac1 AC: C side <= 1", "194-196", "TTATTTT", flank([site("AC", 0.0, (10, 1, 0, 0, 0, 0, 0, 0))]),
     {102: {"emit": 1, "flt": 128, "gt": 1, "alt": "C", "gt_enc": 0x24}})
case("AC: both alleles >= 2 (converted reads count: C = c1 + c5 + c7, A = c0 + c4)", "195", "TTATTTT",
     flank([site("AC", 0.0, (1, 0, 0, 0, 1, 1, 0, 1))]), {102: {"emit": 1, "flt": 0, "dp": 1, "qd": 255}})
case("mac1 AG: only counts[0] stands for A (a G2A-strand A is no evidence against G)", "197-199", "TTATTTT",
     flank([site("AG", 0.0, (1, 0, 10, 0, 9, 0, 0, 0))]), {102: {"emit": 1, "flt": 128, "gt": 2}})
case("AG passes with counts[0] = 2", "198", "TTATTTT", flank([site("AG", 0.0, (2, 0, 1, 0, 0, 0, 1, 0))]),
     {102: {"emit": 1, "flt": 0}})
case("mac1 AT: T = c3 + c7", "200-202", "TTATTTT", flank([site("AT", 0.0, (10, 0, 0, 1, 0, 0, 0, 0))]),
     {102: {"emit": 1, "flt": 128, "gt": 3, "alt": "T"}})
case("mac1 CG: G side = c2 + c6 + c4", "203-205", "TTCTTTT", flank([site("CG", 0.0, (0, 10, 0, 0, 1, 0, 0, 0))]),
     {102: {"emit": 1, "flt": 128, "gt": 5, "alt": "G"}})
case("mac1 CT: only counts[3] stands for T", "206-208", "TTCTTTT", flank([site("CT", 0.0, (0, 10, 0, 1, 0, 0, 0, 9))]),
     {102: {"emit": 1, "flt": 128, "gt": 6}})
case("mac1 GT: G = c2 + c6 + c4", "209-211", "TTGTTTT", flank([site("GT", 0.0, (0, 0, 1, 10, 0, 0, 0, 0))]),
     {102: {"emit": 1, "flt": 128, "gt": 8, "alt": "T"}})
# ---- ALT, GT and GL selection for every reference base (:73-84, :272-277, :319-347) -----------------------------------
GP = [-1.0, -2.0, -3.0, -4.0, -5.0, -6.0, -7.0, -8.0, -9.0, -10.0]  # gt_prob[i] = -(i + 1): AA -1 ... TT -10


def gp_call(g, v=-0.25):
    z = list(GP)
    z[G[g]] = v  # the call: the largest value
    return z


case("GL ref A, call AA: hom-ref only", "322-329", "TTATTTT", flank([site("AA", gp=gp_call("AA"), counts=A30)]),
     {102: {"emit": 1, "all_positions": 1, "gt": 0, "alt": "", "gt_enc": 0x22, "n_gl": 1, "gl": [-0.25]}}, all_positions=True)
case("GL ref A, call AC: [AA, AC, CC]", "330-346", "TTATTTT", flank([site("AC", gp=gp_call("AC"), counts=(15, 15, 0, 0, 0, 0, 0, 0))]),
     {102: {"emit": 1, "gt": 1, "alt": "C", "gt_enc": 0x24, "n_gl": 3, "gl": [-1.0, -0.25, -5.0]}})
case("GL ref C, call AT: two ALT alleles [CC, AC, AA, CT, TT]", "330-346", "TTCTTTT",
     flank([site("AT", gp=gp_call("AT"), counts=(15, 0, 0, 15, 0, 0, 0, 0))]),
     {102: {"emit": 1, "gt": 3, "alt": "AT", "gt_enc": 0x48, "n_gl": 5, "gl": [-5.0, -2.0, -1.0, -7.0, -10.0]}})
case("GL ref G, call GG hom-ref (written: only AA / TT hom-ref are skipped)", "139,322-327", "TTGTTTT",
     flank([site("GG", gp=gp_call("GG"), counts=G30)]), {102: {"emit": 1, "gt": 7, "alt": "", "gt_enc": 0x22, "n_gl": 1, "gl": [-0.25]}})
case("GL ref T, call CT: ALT index below the reference index [TT, CT, CC]", "333-335", "TTTTTTT",
     flank([site("CT", gp=gp_call("CT"), counts=(0, 15, 0, 15, 0, 0, 0, 0))]),
     {102: {"emit": 1, "gt": 6, "alt": "C", "gt_enc": 0x24, "n_gl": 3, "gl": [-10.0, -0.25, -5.0]}})
case("GL on reference N: no reference entries [-99.999, AA, CC]", "327,332", "TTNTTTT",
     flank([site("AC", gp=gp_call("AC"), counts=(15, 15, 0, 0, 0, 0, 0, 0))]),
     {102: {"emit": 1, "gt": 1, "ref_code": 0, "alt": "AC", "gt_enc": 0x48, "n_gl": 3, "gl": [-99.999, -1.0, -5.0], "cx_ref": "TTNNN"}})
case("GL floor at -99.999", "325,337,344", "TTATTTT",
     flank([site("AC", gp=[-250.0, 0.0, -3.0, -4.0, -1e9, -6.0, -7.0, -8.0, -9.0, -10.0], counts=(15, 15, 0, 0, 0, 0, 0, 0))]),
     {102: {"emit": 1, "n_gl": 3, "gl": [-99.999, 0.0, -99.999], "phred": 255}})
# ---- hom-ref skip rule, dbSNP, -A, region (:85-96, :139, :154-158) ------------------------------------------------------
case("AA on reference A is not written (but is formed: position, genotype, context; QUAL and QD are not kept)", "139", "TTATTTT",
     flank([site("AA", 0.0, A30)]), {102: {"emit": 0, "pos": 102, "gt": 0, "phred": 0, "qd": 0, "flt": 0, "n_gl": 0, "alt": "", "cx_ref": "TTATT"}})  # unwritten: no QUAL / QD kept
case("AA on reference A with a dbSNP fq_mask site (rs_found = 3) is written", "139", "TTATTTT", flank([site("AA", 0.0, A30)]),
     {102: {"emit": 1, "gt": 0, "alt": "", "gt_enc": 0x22}}, dbsnp=[0, 0, 3, 0, 0])
case("rs_found = 1 does not force it", "139", "TTATTTT", flank([site("AA", 0.0, A30)]), {102: {"emit": 0}}, dbsnp=[0, 0, 1, 0, 0])
case("TT on reference T with -A is written", "139", "TTTTTTT", flank([site("TT", 0.0, T30)]),
     {100: {"emit": 1}, 102: {"emit": 1, "gt": 9, "gt_enc": 0x22}, 104: {"emit": 1}}, all_positions=True)
case("region clip", "154-158", "TTCTTTT", flank([site("CC", 0.0, C30)], left=("CC", "CC"), right=("CC", "CC")),
     {100: {"emit": 0}, 101: {"emit": 1}, 102: {"emit": 1}, 103: {"emit": 0}, 104: {"emit": 0}}, reg=(101, 102))
case("an uncovered position makes no record and reads as N in its neighbours' context", "555-560,580", "TTCTTTT",
     [site("TT", 0.0, T30), NONE, site("CC", 0.0, C30), NONE, site("TT", 0.0, T30)],
     {101: None, 102: {"emit": 1, "cx_gt": "TNCNT", "cg": "?"}, 103: None})
# ---- CpG status (:231-266) --------------------------------------------------------------------------------------------
case("CC followed by GG: both CG", "231-233", "TTCGTTTT", flank([site("CC", 0.0, C30), site("GG", 0.0, G30)]),
     {102: {"cg": "C", "cx_gt": "TTCGT"}, 103: {"cg": "C", "cx_gt": "TCGTT"}})
case("CC then a G-carrying het: H", "234-238", "TTCATTTT", flank([site("CC", 0.0, C30), site("AG", 0.0, (15, 0, 15, 0, 0, 0, 0, 0))]),
     {102: {"cg": "H"}, 103: {"cg": "H"}})  # 103: AG carries G, the call before it (CC) carries C: H (:258-262)
case("CC then no G: N", "234-238", "TTCATTTT", flank([site("CC", 0.0, C30), site("AA", 0.0, A30)]), {102: {"cg": "N"}})
case("CC then nothing called: ?", "239-240", "TTCTTTTT", [site("TT", 0.0, T30), site("TT", 0.0, T30), site("CC", 0.0, C30), NONE,
     site("TT", 0.0, T30), site("TT", 0.0, T30)], {102: {"cg": "?"}})
case("GG after a C-carrying call: H; after none: N; after nothing: ?", "241-249", "TCGAGTGTT",
     [site("CT", 0.0, (0, 15, 0, 15, 0, 0, 0, 0)), site("GG", 0.0, G30), site("AA", 0.0, A30), site("GG", 0.0, G30), NONE,
      site("GG", 0.0, G30), site("TT", 0.0, T30)],
     {101: {"cg": "H"}, 103: {"cg": "N"}, 105: {"cg": "?"}}, x=100)
case("a C-carrying het (AC) before GT: H; AG after nothing: '.' (not '?')", "250-266", "TTATAGTT",
     [site("TT", 0.0, T30), site("TT", 0.0, T30), site("AC", 0.0, (15, 15, 0, 0, 0, 0, 0, 0)), site("GT", 0.0, (0, 0, 15, 15, 0, 0, 0, 0)),
      NONE, site("AG", 0.0, (15, 0, 15, 0, 0, 0, 0, 0))],
     {102: {"cg": "H"}, 103: {"cg": "H"}, 105: {"cg": "."}})
case("the CG het takes the C branch", "250", "TTCGTTTT", flank([site("CG", 0.0, (0, 15, 15, 0, 0, 0, 0, 0)), site("GG", 0.0, G30)]),
     {102: {"cg": "H"}, 103: {"cg": "H"}})  # 103: GG after CG (cflag) -> H
case("AA: '.'", "231-266", "TTATTTT", flank([site("AA", 0.0, A30)]), {102: {"cg": "."}}, all_positions=True)
# ---- the two positions flushed at the end of a block (:536-546) see the last genotype repeated ------------------------
case("flush: last two positions", "540-543", "AACGCGA",
     [site("AA", 0.0, A30), site("AA", 0.0, A30), site("CC", 0.0, C30), site("GG", 0.0, G30), site("CC", 0.0, C30)],
     {102: {"cx_gt": "AACGC", "cg": "C"}, 103: {"cx_gt": "ACGCC", "cg": "C", "cx_ref": "ACGCG"},
      104: {"cx_gt": "CGCCC", "cg": "N", "cx_ref": "CGCGA"}})  # 104: CC "followed by" its own repeat CC: no G -> N
case("flush: a GG at the block end after CC is CG; a one-position block sees itself either side", "540-543", "GAT",
     [site("GG", 0.0, G30)], {100: {"cx_gt": "NNGGG", "cg": "?", "cx_ref": "NNGAT", "emit": 1}})
# ---- the reference's strncpy of the context: an N blanks everything after it (:570-577) -------------------------------
case("N in the reference window", "570-577", "ACGTNACGT", [site("AA", 0.0, A30), site("CC", 0.0, C30), site("GG", 0.0, G30),
     site("TT", 0.0, T30), site("AA", 0.0, A30), site("AA", 0.0, A30), site("CC", 0.0, C30)],
     {101: {"cx_ref": "NACGT"}, 102: {"cx_ref": "ACGTN"}, 103: {"cx_ref": "CGTNN"}, 104: {"cx_ref": "GTNNN", "ref_code": 0},
      105: {"cx_ref": "TNNNN", "ref_code": 0, "emit": 1}, 106: {"cx_ref": "NNNNN", "ref_code": 0}}, all_positions=True)

out = {"_doc": __doc__, "genotypes": list(G), "cases": cases}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "vcf_kav.json"), "w"), indent=1)
print("%d cases, %d checked records" % (len(cases), sum(len(c["expect"]) for c in cases)))
