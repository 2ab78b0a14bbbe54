"""py_prep.py — pure-Python restatement of the reference's read pre-processing, the checker of csrc/prep.c.

*** TEST INFRASTRUCTURE, NOT PRODUCT ***.  Follows the reference statement by statement on Python lists:
  trim_read         src/read_utils.c:13-26      (incl. the right trim copying the base of sp[k1])
  trim_soft_clips   src/al_utils.c:122-162
  handle_overlap    src/al_utils.c:164-318
  indel normalisation  src/process_template.c:62-108
  get_al_qual       src/al_utils.c:19-35        (incl. the sq[k] indexing)
  meth_profile      src/meth_profile.c:48-77    (the non-CpG read profile, with the positions process_template.c:76-108 tracks)
Pinning: like the rest of the oracle, nothing in this image can run the reference's own code for this stage (it needs
the gt/ containers built against htslib-dependent headers); the restatement is pinned by the hand-worked cases of
tests/test_prep.py, each derived from the cited lines.
A template is a dict: pos [fwd, rev], span [2], reads [list of ints or None, ...], misms [list of [type, position, size]
per read], mapq, orientation, bs_strand.  Types: 0 MISMS, 1 INS (CIGAR D), 2 DEL (CIGAR I), 3 SOFT.
"""
import copy

FLT_QUAL = 63
INS, DEL, SOFT = 1, 2, 3


class PrepError(Exception):
    pass


def trim_read(rd, left, right):
    if rd is None:
        return
    rl = len(rd)
    if rl > 0:
        for k1 in range(min(left, rl)):
            rd[k1] = (rd[k1] & 3) | (FLT_QUAL << 2)
        for k1 in range(min(right, rl)):
            rd[rl - k1 - 1] = (rd[k1] & 3) | (FLT_QUAL << 2)


def _left_trim(rd, l):
    if l > 0:
        if l >= len(rd):
            del rd[:]
        else:
            del rd[:l]


def _right_trim(rd, l):
    if l > 0:
        if l >= len(rd):
            del rd[:]
        else:
            del rd[len(rd) - l :]


def trim_soft_clips(al, st, trim_left=None, trim_right=None):
    for k in range(2):
        rd = al["reads"][k]
        if rd is None:
            continue
        rl = len(rd)
        if rl == 0:
            continue
        ms = al["misms"][k]
        num = len(ms)
        nclip, adj = 0, 0
        for z in range(num):
            m = ms[z]
            if m[0] == SOFT:
                if z and z != num - 1:
                    raise PrepError("Soft clip not at extremity of read")
                nclip += 1
                if not m[1]:
                    if m[2] >= rl:
                        raise PrepError("Illegal soft clip")
                    adj = m[2]
                    st["base_clip"] += adj
                    if trim_left is not None:
                        trim_left[k] = adj  # :143
                    _left_trim(rd, adj)
                else:
                    if m[1] + m[2] != rl:
                        raise PrepError("Illegal soft clip")
                    _right_trim(rd, m[2])
                    if trim_right is not None:
                        trim_right[k] = m[2]  # :149
                    st["base_clip"] += m[2]
            elif nclip:
                m[1] -= adj
                ms[z - nclip] = list(m)
        if nclip:
            del ms[num - nclip :]


def handle_overlap(al, st, trim_left=None, trim_right=None):
    rdl = [len(r) if r is not None else 0 for r in al["reads"]]
    if not (rdl[0] > 0 and rdl[1] > 0):
        return
    pos, span = al["pos"], al["span"]
    if pos[0] <= pos[1]:
        overlap = span[0] - pos[1] + pos[0]
        rev = False
    else:
        overlap = span[1] + pos[1] - pos[0]
        rev = True
    overlap = ((overlap + 2**31) % 2**32) - 2**31  # int32_t
    if not (pos[0] + span[0] >= pos[1]):
        return
    if span[0] > span[1]:
        tr = 1
    elif span[0] < span[1]:
        tr = 0
    else:
        tot = [0, 0]
        for k in range(2):
            n = 0
            for b in al["reads"][k]:
                q = b >> 2
                if q != FLT_QUAL:
                    tot[k] += q
                    n += 1
            tot[k] = tot[k] // n if n > 0 else 0
        tr = 0 if tot[0] <= tot[1] else 1
    right = (rev and tr) or not (rev or tr)
    if not right:
        if tr:
            pos[1] = (pos[1] + overlap) % 2**32
        else:
            pos[0] = (pos[0] + overlap) % 2**32
    rd, ms = al["reads"][tr], al["misms"][tr]
    num = len(ms)
    u32 = lambda v: v % 2**32
    if not num:
        (_right_trim if right else _left_trim)(rd, u32(overlap))
    else:
        trimmed = False
        if right:
            xx = u32(span[tr] - overlap)
            adj = 0
            z = 0
            while z < num:
                m = ms[z]
                if m[1] + adj >= xx:
                    trim = rdl[tr] - xx + adj
                    _right_trim(rd, u32(trim))
                    num = z
                    trimmed = True
                    break
                if m[0] == INS:
                    if m[1] + adj + m[2] >= xx:
                        trim = rdl[tr] - m[1]
                        m[2] = u32(xx - (m[1] + adj))
                        _right_trim(rd, u32(trim))
                        num = z + 1
                        trimmed = True
                    adj += m[2]
                elif m[0] == DEL:
                    adj -= m[2]
                z += 1
            if not trimmed:
                _right_trim(rd, u32(overlap))
        else:
            xx = u32(overlap)
            adj = 0
            for z in range(num):
                m = ms[z]
                if m[1] + adj >= xx:
                    trim = u32(overlap - adj)
                    _left_trim(rd, trim)
                    trimmed = True
                    if z:
                        for z1 in range(z, num):
                            ms[z1][1] = u32(ms[z1][1] - trim)
                            ms[z1 - z], ms[z1] = ms[z1], ms[z1 - z]
                        num -= z
                    else:
                        for z1 in range(num):
                            ms[z1][1] = u32(ms[z1][1] - trim)
                    break
                if m[0] == INS:
                    if m[1] + adj + m[2] >= xx:
                        m[2] = u32(m[1] + m[2] + adj - xx)
                        trim = m[1]
                        _left_trim(rd, trim)
                        trimmed = True
                        z2 = z if m[2] else z + 1
                        for z1 in range(z2, num):
                            ms[z1][1] = u32(ms[z1][1] - trim)
                            if z2:
                                ms[z1 - z2], ms[z1] = ms[z1], ms[z1 - z2]
                        num -= z2
                        break
                    adj += m[2]
                elif m[0] == DEL:
                    adj -= m[2]
            if not trimmed:
                _left_trim(rd, u32(overlap - adj))
                num = 0
        del ms[num:]
    rdl1 = [len(r) if r is not None else 0 for r in al["reads"]]
    st["base_overlap"] += rdl[0] - rdl1[0] + rdl[1] - rdl1[1]
    if trim_left is not None:  # :309-313
        if right:
            trim_right[tr] += rdl[tr] - rdl1[tr]
        else:
            trim_left[tr] += rdl[tr] - rdl1[tr]


def normalise(rd, ms, orig=None):
    """src/process_template.c:62-108: returns the read with deletions padded (byte 0) and insertions removed; `orig` (the
    position of every base in the original read) is edited alongside (-1 for padding)."""
    sp = list(rd)
    rl = len(sp)
    npad = sum(m[2] for m in ms if m[0] == INS)
    sp += [0] * npad  # gt_vector_reserve_additional
    if orig is not None:
        orig += [0] * npad
    adj = 0
    for m in ms:
        ix1 = m[1] + adj
        # the reference memmoves rl + adj - ix1 [- size] bytes here: a negative count (an indel the overlap trimming left
        # beyond the end of the read) is undefined behaviour there; csrc/prep.c reports it, so does this restatement
        if ix1 < 0 or (m[0] == INS and ix1 > rl + adj) or (m[0] == DEL and ix1 + m[2] > rl + adj):
            raise PrepError("indel beyond the read")
        if m[0] == INS:
            sp[ix1 + m[2] : ix1 + m[2] + (rl + adj - ix1)] = sp[ix1 : rl + adj]
            if orig is not None:
                orig[ix1 + m[2] : ix1 + m[2] + (rl + adj - ix1)] = orig[ix1 : rl + adj]
            for k1 in range(m[2]):
                sp[ix1 + k1] = 0
                if orig is not None:
                    orig[ix1 + k1] = -1
            adj += m[2]
        elif m[0] == DEL:
            n = rl + adj - ix1 - m[2]
            sp[ix1 : ix1 + n] = sp[ix1 + m[2] : ix1 + m[2] + n]
            if orig is not None:
                orig[ix1 : ix1 + n] = orig[ix1 + m[2] : ix1 + m[2] + n]
            adj -= m[2]
    if orig is not None:
        del orig[rl + adj :]
    return sp[: rl + adj]


# src/meth_profile.c:14-23 — indexed by (previous reference code << 3 | current): 4 = "CA, CC or CT", 8 = "AG, GG or TG"
RTAB = [0] * 64
for _prev, _cur, _v in ((2, 1, 4), (2, 2, 4), (2, 4, 4), (1, 3, 8), (3, 3, 8), (4, 3, 8)):
    RTAB[_prev << 3 | _cur] = _v


def flt_tab(bs_strand, byte, min_q=20):
    """par->work.flt_tab (src/init_param.c:57-70): entries exist for qualities MIN_QUAL .. FLT_QUAL - 1 only."""
    q = byte >> 2
    if q < min_q or q >= FLT_QUAL:
        return 0
    return ((11, 6, 10, 7), (11, 4, 10, 5), (9, 6, 8, 7))[bs_strand][byte & 3]


class Profile:
    """bs_stats.meth_profile: a gt_vector of meth_cts, 256 elements allocated and zeroed at the start (src/stats.c:305-309)."""

    def __init__(self, allocated=4096):
        self.mem = [[0, 0, 0, 0] for _ in range(allocated)]
        self.used = 0


def meth_profile(prof, al, x, ref, orig_pos, max_pos):
    """src/meth_profile.c:48-77.  `al`: a prepared template (reads normalised), `ref`: codes of x .. (work->ref1)."""
    if max_pos + 1 > prof.used:  # gt_vector_reserve(.., true): everything from the old end on is cleared (gt_vector.c:34-37)
        for e in prof.mem[prof.used :]:
            e[:] = [0, 0, 0, 0]
        prof.used = max_pos + 1
    for k in range(2):
        sp = al["reads"][k]
        if not sp:
            continue
        pos = al["pos"][1] if k else al["pos"][0]
        r = pos - x
        if pos > x:
            state = (ref[r - 1] << 3) | ref[r]
            r += 1
        else:
            state = 0
        mask = RTAB[state]
        for j in range(len(sp)):
            xx = flt_tab(al["bs_strand"], sp[j])
            cts = prof.mem[1 + orig_pos[k][j]]
            mask1 = (xx & mask) >> 1
            state = ((state << 3) | ref[r]) & 63 if pos >= x else 0
            r += 1
            mask = RTAB[state]
            cts[xx & 3] += (((xx & mask) | mask1) >> 2) & 1


def prepare(templates, left_trim=(0, 0), right_trim=(0, 0), min_qual=20, profile=None, x=None, ref=None):
    """process_template_vector's per-template loop (src/process_template.c:36-111) -> (prepared templates, stats); with
    `profile` (a Profile), block start `x` and the block's reference codes `ref`, also what the mprof thread adds."""
    st = {"base_none": 0, "base_trim": 0, "base_clip": 0, "base_overlap": 0, "base_lowqual": 0, "reads": 0, "read_bases": 0}
    out = []
    for al0 in templates:
        al = copy.deepcopy(al0)
        msk = 0 if al["orientation"] == 0 else 1
        if left_trim[0] or right_trim[0]:
            trim_read(al["reads"][0 ^ msk], left_trim[0], right_trim[0])
        if left_trim[1] or right_trim[1]:
            trim_read(al["reads"][1 ^ msk], left_trim[1], right_trim[1])
        tl, tr_ = [0, 0], [0, 0]
        trim_soft_clips(al, st, tl, tr_)
        handle_overlap(al, st, tl, tr_)
        reads = []
        origs, max_pos = [[], []], 0
        for k in range(2):
            rd = al["reads"][k]
            if rd is None:
                reads.append([])
                continue
            rl = len(rd)
            if k:  # :76-87
                posx = rl + tr_[k] - 1
                origs[k] = [posx - k1 for k1 in range(rl)]
                mpos = posx
            else:
                posx = tl[k]
                origs[k] = [posx + k1 for k1 in range(rl)]
                mpos = posx + rl
            max_pos = max(max_pos, mpos)
            for c in rd:
                q = c >> 2
                if q == FLT_QUAL:
                    st["base_trim"] += 1
                elif q < min_qual:
                    st["base_lowqual"] += 1
                else:
                    st["base_none"] += 1
            st["reads"] += 1
            st["read_bases"] += len(rd)
            reads.append(normalise(rd, al["misms"][k], origs[k]))
        out.append({"pos": list(al["pos"]), "reads": reads, "mapq": al["mapq"], "orientation": al["orientation"], "bs_strand": al["bs_strand"]})
        if profile is not None:
            meth_profile(profile, out[-1], x, ref, origs, max_pos)
    return out, st


def get_al_qual(al):
    qual = n = 0
    for k in range(2):
        rd = al["reads"][k]
        if rd is not None:
            for _ in range(len(rd)):
                q = rd[k] >> 2  # sq[k]: the reference's indexing
                if q != FLT_QUAL:
                    qual += q
                    n += 1
    return qual // n if n > 0 else 0
