#!/usr/bin/env python3
"""Two contexts driven by two host threads (one context = one stream-ordered sequence): block k+1's upload and
kernels overlap block k's copy-out, both PCIe directions stay busy.  usage: python tools/bench_two_contexts.py [sites] [cov]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bs_call_amd as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
cov = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tpl, seq = B.synth_reads_host(88172645463325252, 1000, n, cov)
x, y = 998, int((tpl["pos"] + tpl["len"]).max()) - 1
nn = y - x + 1
ref = B.synth_ref_host(88172645463325252, x, nn + 2)
REPS = 6


def worker(c, bufs, out_counts, k):
    p_tpl, p_seq, p_ref, p_rec = bufs
    for _ in range(REPS):
        recs = c.block_records(p_tpl.array, p_seq.array, x, y, p_ref.array, out=p_rec.array)
    out_counts[k] = len(recs)


for nctx in (1, 2, 3):
    ctxs = [B.SiteCaller() for _ in range(nctx)]
    bufs = []
    for c in ctxs:
        b = (B.PinnedBuffer(len(tpl), tpl.dtype), B.PinnedBuffer(len(seq), np.uint8), B.PinnedBuffer(nn + 2, np.uint8),
             B.PinnedBuffer(nn, B.VCF_REC))
        b[0].array[:], b[1].array[:], b[2].array[:] = tpl, seq, ref
        c.block_records(b[0].array, b[1].array, x, y, b[2].array, out=b[3].array)  # warm-up: allocations
        bufs.append(b)
    counts = [0] * nctx
    th = [threading.Thread(target=worker, args=(ctxs[k], bufs[k], counts, k)) for k in range(nctx)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print("%d context(s): %d blocks of %d positions in %.1f ms -> %.1f M positions/s (reads in, packed records out)" % (
        nctx, nctx * REPS, nn, dt * 1e3, nctx * REPS * nn / dt / 1e6))
    assert len(set(counts)) == 1
    for c in ctxs:
        c.close()
