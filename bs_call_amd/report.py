"""The JSON report of a run: `bsc_report_json` (csrc/report.c), the text the reference's output_stats() writes
(src/stats.c:19-298), from the statistics block, the read-level counters and the per-contig totals."""
import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .abi import SITE_STATS

TOTAL_FIELDS = ("snps", "indels", "multi", "dbSNP_sites", "dbSNP_var", "CpG_ref", "CpG_nonref")
COV_CAP = 4096


def render_json(total, under_conv: float = 0.01, over_conv: float = 0.05, mapq_thresh: int = 20, min_qual: int = 20,
                date: Optional[Tuple[int, int, int]] = None, have_dbsnp: bool = False, filter_cts: Sequence[int] = (),
                filter_bases: Sequence[int] = (), base_filter: Sequence[int] = (), gc=None, read_profile=None,
                contigs: Sequence[Tuple[str, np.ndarray]] = ()) -> str:
    """`total`: a SITE_STATS record (SiteCaller.site_stats()); `date` (day, month, year) or None for today; `gc` a
    (4096, 101) uint64 array or None; `read_profile` an (n, 4) uint64 array (element 0 is never reported) or None;
    `contigs` = [(name, (7, 2) totals as SiteCaller.site_totals() differences)]."""
    L = _lib.load()
    tot = np.ascontiguousarray(np.asarray(total, dtype=SITE_STATS).reshape(1))
    r = _lib.Report()
    r.under_conv, r.over_conv, r.mapq_thresh, r.min_qual = under_conv, over_conv, mapq_thresh, min_qual
    if date is not None:
        r.day, r.month, r.year = date
    r.have_dbsnp = 1 if have_dbsnp else 0
    for name, vals, n in (("filter_cts", filter_cts, 15), ("filter_bases", filter_bases, 15), ("base_filter", base_filter, 5)):
        if len(vals) > n:
            raise ValueError("%s has at most %d entries" % (name, n))
        arr = getattr(r, name)
        for i, v in enumerate(vals):
            arr[i] = int(v)
    r.total = tot.ctypes.data
    keep = [tot]
    if gc is not None:
        g = np.ascontiguousarray(gc, dtype=np.uint64)
        if g.shape != (COV_CAP, 101):
            raise ValueError("gc must be (%d, 101)" % COV_CAP)
        r.gc = g.ctypes.data
        keep.append(g)
    if read_profile is not None:
        rp = np.ascontiguousarray(read_profile, dtype=np.uint64)
        if rp.ndim != 2 or rp.shape[1] != 4:
            raise ValueError("read_profile must be (n, 4)")
        r.read_profile = rp.ctypes.data
        r.n_read_profile = rp.shape[0]
        keep.append(rp)
    ct = (_lib.ContigTotals * max(1, len(contigs)))()
    for i, (name, t) in enumerate(contigs):
        t = np.asarray(t, dtype=np.uint64).reshape(7, 2)
        ct[i].name = name.encode()
        for k, f in enumerate(TOTAL_FIELDS):
            getattr(ct[i], f)[0], getattr(ct[i], f)[1] = int(t[k, 0]), int(t[k, 1])
    r.contigs = ct
    r.n_contigs = len(contigs)
    need = L.bsc_report_json(C.byref(r), None, 0)
    if need < 0:
        raise ValueError("bsc_report_json: bad argument")
    buf = C.create_string_buffer(need + 1)
    got = L.bsc_report_json(C.byref(r), buf, need + 1)
    assert got == need
    return buf.raw[:need].decode()
