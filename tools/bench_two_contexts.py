#!/usr/bin/env python3
"""Host buffers in, packed records out, PCIe included (never bench.py's `value`):
  (a) bsc_block_records, one context, one block after the other;
  (b) N contexts driven by N host threads, each running (a) — the threads fall into step (all uploading, then all copying
      out), so the two directions of the link are rarely busy together;
  (c) the pipeline a host that flattens block k + 1 while block k is in flight would run, from ONE host thread: two contexts
      alternating, bsc_block_records_submit_inplace(k + 1) before bsc_block_records_fetch(k) — block k + 1's upload and
      kernels overlap block k's copy-out.
usage: python tools/bench_two_contexts.py [sites] [cov]"""
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


def buffers():
    b = (B.PinnedBuffer(len(tpl), tpl.dtype), B.PinnedBuffer(len(seq), np.uint8), B.PinnedBuffer(nn + 2, np.uint8), B.PinnedBuffer(nn, B.VCF_REC))
    b[0].array[:], b[1].array[:], b[2].array[:] = tpl, seq, ref
    return b


def worker(c, bufs, out_counts, k):
    p_tpl, p_seq, p_ref, p_rec = bufs
    for _ in range(REPS):
        recs = c.block_records(p_tpl.array, p_seq.array, x, y, p_ref.array, out=p_rec.array)
    out_counts[k] = len(recs)


for nctx in (() if os.environ.get('ONLY_PIPELINE') else (1, 2, 3)):
    ctxs = [B.SiteCaller() for _ in range(nctx)]
    bufs = [buffers() for _ in ctxs]
    for c, b in zip(ctxs, bufs):
        c.block_records(b[0].array, b[1].array, x, y, b[2].array, out=b[3].array)  # warm-up: allocations
    counts = [0] * nctx
    th = [threading.Thread(target=worker, args=(ctxs[k], bufs[k], counts, k)) for k in range(nctx)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print("%d context(s), one thread each: %d blocks of %d positions in %.1f ms -> %.1f M positions/s" % (
        nctx, nctx * REPS, nn, dt * 1e3, nctx * REPS * nn / dt / 1e6), flush=True)
    assert len(set(counts)) == 1
    for c in ctxs:
        c.close()

ctxs = [B.SiteCaller(), B.SiteCaller()]
bufs = [buffers(), buffers()]
for c, b in zip(ctxs, bufs):
    c.block_records(b[0].array, b[1].array, x, y, b[2].array, out=b[3].array)
NB = 12
t0 = time.perf_counter()
lens = []
for k in range(NB):
    b = bufs[k & 1]
    ctxs[k & 1].block_records_submit(b[0].array, b[1].array, x, y, b[2].array, b[3].array, inplace=True)
    if k:
        lens.append(len(ctxs[(k - 1) & 1].block_records_fetch()))
lens.append(len(ctxs[(NB - 1) & 1].block_records_fetch()))
dt = time.perf_counter() - t0
assert len(set(lens)) == 1
print("pipeline, one host thread, two contexts alternating (submit_inplace k+1 before fetch k): %d blocks in %.1f ms -> %.1f M positions/s" % (
    NB, dt * 1e3, NB * nn / dt / 1e6))
for c in ctxs:
    c.close()

# the same with the BCF stream coming back instead of the packed records (bsc_block_bcf; bsc_block_bcf_submit_inplace / _fetch)
c = B.SiteCaller()
pb = bufs[0]
blob, n_rec = c.block_bcf(pb[0].array, pb[1].array, x, y, pb[2].array, 0)
outs = [B.PinnedBuffer(len(blob) + 4096, np.uint8), B.PinnedBuffer(len(blob) + 4096, np.uint8)]
t0 = time.perf_counter()
for _ in range(REPS):
    c.block_bcf_submit(pb[0].array, pb[1].array, x, y, pb[2].array, 0, outs[0].array, inplace=True)
    g, nr_ = c.block_bcf_fetch()
dt = time.perf_counter() - t0
assert nr_ == n_rec and len(g) == len(blob)
print("BCF stream back (%.1f bytes per position), one context: %d blocks in %.1f ms -> %.1f M positions/s" % (len(blob) / nn, REPS, dt * 1e3, REPS * nn / dt / 1e6))
ctxs = [c, B.SiteCaller()]
ctxs[1].block_bcf(pb[0].array, pb[1].array, x, y, pb[2].array, 0)
t0 = time.perf_counter()
for k in range(NB):
    ctxs[k & 1].block_bcf_submit(pb[0].array, pb[1].array, x, y, pb[2].array, 0, outs[k & 1].array, inplace=True)
    if k:
        g, nr_ = ctxs[(k - 1) & 1].block_bcf_fetch()
g, nr_ = ctxs[(NB - 1) & 1].block_bcf_fetch()
dt = time.perf_counter() - t0
assert nr_ == n_rec and g.tobytes() == blob
print("BCF stream back, one host thread, two contexts alternating (submit_inplace k+1 before fetch k): %d blocks in %.1f ms -> %.1f M positions/s" % (
    NB, dt * 1e3, NB * nn / dt / 1e6))
for cc in ctxs:
    cc.close()
