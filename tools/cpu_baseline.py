#!/usr/bin/env python3
"""The three CPU timings SURVEY.md section 8d asks for, on this host's cores, with the oracle's libm flavour (= the
reference's arithmetic): (i) calc_gt_prob only, (ii) the full per-site path (summary + model + Fisher + 200-byte record),
(iii) the pile-up accumulate; single thread and all hardware threads.   usage: python tools/cpu_baseline.py [sites] [cov]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bs_call_amd as B
from oracle import loader as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
cov = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cores = os.cpu_count() or 1
tb = O.Tables()
L = O.lib()
pile, ref = B.synth_pileup_host(88172645463325254, 0, n, cov)
out = np.zeros(n, dtype=B.GT_METH)
skip = np.zeros(n, dtype=np.uint8)
out[:] = out


def t(f):
    t0 = time.perf_counter()
    f()
    return time.perf_counter() - t0


# (ii) full per-site path
full1 = t(lambda: L.orc_call_sites(pile.ctypes.data, ref.ctypes.data, n, tb.ptr, out.ctypes.data, skip.ctypes.data, O.LIBM, 1))
fullT = t(lambda: L.orc_call_sites(pile.ctypes.data, ref.ctypes.data, n, tb.ptr, out.ctypes.data, skip.ctypes.data, O.LIBM, -cores))
fullS = t(lambda: L.orc_call_sites(pile.ctypes.data, ref.ctypes.data, n, tb.ptr, out.ctypes.data, skip.ctypes.data, O.LIBM, min(cores, 64)))
# (i) model only, on the prepared records of the covered sites
gt = out[skip == 0].copy()
rf = ref[skip == 0].copy()
model1 = t(lambda: L.orc_calc_gt_prob_array(gt.ctypes.data, rf.ctypes.data, len(gt), tb.ptr, O.LIBM))
# (iii) accumulate
m = min(n, 2_000_000)
tpl, seq = B.synth_reads_host(88172645463325254, 1000, m, cov)
x, y = 998, int((tpl["pos"] + tpl["len"]).max()) - 1
pl = np.zeros(y - x + 1, dtype=B.PILEUP)
acc1 = t(lambda: L.orc_accumulate(tpl.ctypes.data, len(tpl), seq.ctypes.data, x, y, 20, pl.ctypes.data))
print("host: %d hardware threads; %d positions at %dx" % (cores, n, cov))
print("(i)   calc_gt_prob only, 1 thread        : %.2f M positions/s (%.0f ns/site)" % (len(gt) / model1 / 1e6, model1 / len(gt) * 1e9))
print("(ii)  full per-site path, 1 thread       : %.2f M positions/s" % (n / full1 / 1e6))
print("(ii)  full per-site path, %3d threads     : %.1f M positions/s (contiguous ranges)" % (cores, n / fullT / 1e6))
print("(ii)  full per-site path, %3d threads     : %.1f M positions/s (the reference's interleaved striding)" % (min(cores, 64), n / fullS / 1e6))
print("(iii) accumulate, 1 thread (serial in the reference): %.1f M bases/s = %.2f M positions/s" % (len(seq) / acc1 / 1e6, (y - x + 1) / acc1 / 1e6))
