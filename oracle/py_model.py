"""Independent pure-Python restatement of calc_gt_prob / get_Z / fisher, written from the likelihood-term MATRIX of
SURVEY.md appendix A rather than from the reference's statement list (which oracle/orc_model.inc follows).

TEST INFRASTRUCTURE (small cases only: pure-Python loops).  Purpose: two restatements with different structure that
agree bit for bit on random inputs make a transcription slip in either of them unlikely.  math.log / math.exp call
the platform libm, i.e. the same functions the reference links; CPython's math.lgamma is its own Lanczos code, so libm's
lgamma is called through ctypes.
"""
import ctypes
import ctypes.util
import math

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.lgamma.restype = ctypes.c_double
_libm.lgamma.argtypes = [ctypes.c_double]

LN10 = 2.30258509299404568402  # LOG10, include/bs_call.h:36
GENOTYPES = ("AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT")
BASES = "ACGT"


def q_prob(q):
    """fill_base_prob_table, src/genotype_model.c:10-21 -> (k, ln_k, ln_k_half, ln_k_one)"""
    e = math.exp(-0.1 * float(q) * LN10)
    if e > 0.5:
        e = 0.5
    k = e / (3.0 - 4.0 * e)
    return k, math.log(k), math.log(0.5 + k), math.log(1.0 + k)


def get_z(x1, x2, k1, k2, l, t):
    """src/genotype_model.c:23-42"""
    lpt, lmt = l + t, l - t
    d = (x1 + x2) * lmt
    out = []
    for a, b in ((lpt + 2.0 * k2, 2.0 - lpt + 2.0 * k1), (2.0 + lpt + 4.0 * k2, 2.0 - lpt + 4.0 * k1),
                 (lpt + 4.0 * k2, 2.0 - lpt + 4.0 * k1)):
        s = (x1 * a - x2 * b) / d
        s = -1.0 if s < -1.0 else (1.0 if s > 1.0 else s)
        out.append(0.5 * (lmt * s + 2.0 - lpt))
    return out


def _term_kinds(c):
    """Appendix A, one row: genotype name -> kind of class c's term ('one', 'half', 'k', 'za', 'zb', 'zc')."""
    kinds = {}
    base = BASES[c & 3]
    for g in GENOTYPES:
        if c < 4:  # non-informative base: 1+k for the homozygote, 1/2+k for hets carrying it, k otherwise
            kinds[g] = "one" if g == base * 2 else ("half" if base in g else "k")
    if c >= 4:
        z = {
            4: {"AA": "one", "AC": "half", "AT": "half", "AG": "za", "GG": "zb", "CG": "zc", "GT": "zc"},
            5: {"CC": "za", "CT": "zb", "AC": "zc", "CG": "zc"},
            6: {"GG": "za", "AG": "zb", "CG": "zc", "GT": "zc"},
            7: {"TT": "one", "AT": "half", "GT": "half", "CC": "za", "CT": "zb", "AC": "zc", "CG": "zc"},
        }[c]
        for g in GENOTYPES:
            kinds[g] = z.get(g, "k")
    return kinds


def calc_gt_prob(counts, quals, rf, under_conv=0.01, over_conv=0.05, ref_bias=2.0):
    """-> (max_gt, [gt_prob x 10]); src/genotype_model.c:44-246 in matrix form."""
    n = [float(c) for c in counts]
    qp = [q_prob(q) for q in quals]
    l, t = 1.0 - under_conv, over_conv
    ll = [0.0] * 10
    if 1 <= rf <= 4:  # prior
        rb = BASES[rf - 1]
        lrb, lrb1 = math.log(ref_bias), math.log(0.5 * (1.0 + ref_bias))
        for i, g in enumerate(GENOTYPES):
            if g == rb * 2:
                ll[i] = lrb
            elif rb in g:
                ll[i] = lrb1
    Z = [-1.0] * 6
    if n[5] + n[7] > 0.0:
        Z[0:3] = get_z(n[5], n[7], qp[5][0], qp[7][0], l, t)
    if n[4] + n[6] > 0.0:
        Z[3:6] = get_z(n[6], n[4], qp[6][0], qp[4][0], l, t)
    zargs = {  # class -> (za, zb, zc) log arguments, evaluated left to right as in the reference
        4: lambda k: (1.0 - 0.5 * Z[4] + k, 1.0 - Z[3] + k, 0.5 * (1.0 - Z[5]) + k),
        5: lambda k: (Z[0] + k, 0.5 * Z[1] + k, 0.5 * Z[2] + k),
        6: lambda k: (Z[3] + k, 0.5 * Z[4] + k, 0.5 * Z[5] + k),
        7: lambda k: (1.0 - Z[0] + k, 1.0 - 0.5 * Z[1] + k, 0.5 * (1.0 - Z[2]) + k),
    }
    for c in range(8):  # one term per class, class order: the order of the reference's += statements
        if not n[c]:
            continue
        k, ln_k, ln_half, ln_one = qp[c]
        val = {"one": n[c] * ln_one, "half": n[c] * ln_half, "k": n[c] * ln_k}
        if c >= 4:
            za, zb, zc = zargs[c](k)
            val.update(za=math.log(za) * n[c], zb=math.log(zb) * n[c], zc=math.log(zc) * n[c])
        kinds = _term_kinds(c)
        for i, g in enumerate(GENOTYPES):
            ll[i] += val[kinds[g]]
    mx, mxi = ll[0], 0
    for i in range(1, 10):
        if ll[i] > mx:
            mx, mxi = ll[i], i
    s = 0.0
    for i in range(10):
        s += math.exp(ll[i] - mx)
    s = math.log(s)
    return mxi, [(ll[i] - mx - s) / LN10 for i in range(10)]


def lfact(x, store):
    return store[x] if x < 256 else _libm.lgamma(float(x + 1))


def fisher(c, store):
    """src/stats_utils.c:25-91 (two-sided Fisher exact p for the 2x2 table c)."""
    c = list(c)
    row = [c[0] + c[1], c[2] + c[3]]
    col = [c[0] + c[2], c[1] + c[3]]
    n = row[0] + row[1]
    if n == 0:
        return 1.0
    delta = float(c[0]) - float(row[0] * col[0]) / float(n)
    knst = lfact(col[0], store) + lfact(col[1], store) + lfact(row[0], store) + lfact(row[1], store) - lfact(n, store)

    def point():
        return math.exp(knst - lfact(c[0], store) - lfact(c[1], store) - lfact(c[2], store) - lfact(c[3], store))

    def walk(dec, inc, steps):  # shrink the `dec` diagonal, grow the `inc` one
        nonlocal l, p
        for i in range(steps):
            l *= float((c[dec[0]] - i) * (c[dec[1]] - i)) / float((c[inc[0]] + i + 1) * (c[inc[1]] + i + 1))
            p += l

    l = point()
    p = l
    main, anti = (0, 3), (1, 2)
    first, second, k = (anti, main, math.ceil(2.0 * delta)) if delta > 0.0 else (main, anti, math.ceil(-2.0 * delta))
    if delta <= 0.0 and not k:
        k = 1
    walk(first, second, min(c[first[0]], c[first[1]]))
    mn = min(c[second[0]], c[second[1]])
    if k <= mn:
        for i in second:
            c[i] -= k
        for i in first:
            c[i] += k
        l = point()
        p += l
        walk(second, first, mn - k)
    return p
