#!/usr/bin/env python3
"""Host-inclusive throughput of bsc_accumulate / bsc_call_block on synthetic reads (L-reads generator).
The host buffers cross PCIe here, so this is the end-to-end figure DESIGN.md quotes next to the HBM-resident
kernel number of bench.py — it is never bench.py's `value`.   usage: python tools/bench_block.py [sites] [cov]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bs_call_amd as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
cov = int(sys.argv[2]) if len(sys.argv) > 2 else 30
t0 = time.time()
tpl, seq = B.synth_reads_host(88172645463325252, 1000, n, cov)
x, y = 998, int((tpl["pos"] + tpl["len"]).max()) - 1
ref = B.synth_ref_host(88172645463325252, x, y - x + 1)
print("generated %d templates, %d bases for %d positions in %.1f s" % (len(tpl), len(seq), y - x + 1, time.time() - t0))
with B.SiteCaller() as c:
    nn = y - x + 1
    pile = np.zeros(nn, dtype=B.PILEUP)  # preallocated and touched: the timing loop measures the library, not page faults
    out = np.zeros(nn, dtype=B.GT_METH)
    skip = np.zeros(nn, dtype=np.uint8)
    c.accumulate(tpl, seq, x, y, out=pile)  # warm-up: allocations
    c.call_block(tpl, seq, x, y, ref, out=out, skip=skip)
    # the same with every host buffer page-locked (bsc_alloc_host): DMA straight to/from the caller's memory
    p_tpl, p_seq, p_ref = B.PinnedBuffer(len(tpl), tpl.dtype), B.PinnedBuffer(len(seq), np.uint8), B.PinnedBuffer(len(ref), np.uint8)
    p_tpl.array[:], p_seq.array[:], p_ref.array[:] = tpl, seq, ref
    p_pile, p_out, p_skip = B.PinnedBuffer(nn, B.PILEUP), B.PinnedBuffer(nn, B.GT_METH), B.PinnedBuffer(nn, np.uint8)
    for name, fn in (("accumulate", lambda: c.accumulate(tpl, seq, x, y, out=pile)),
                     ("call_block", lambda: c.call_block(tpl, seq, x, y, ref, out=out, skip=skip)),
                     ("accumulate/pinned", lambda: c.accumulate(p_tpl.array, p_seq.array, x, y, out=p_pile.array)),
                     ("call_block/pinned", lambda: c.call_block(p_tpl.array, p_seq.array, x, y, p_ref.array,
                                                                out=p_out.array, skip=p_skip.array))):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        t = min(ts)
        print("%-20s %.1f ms  -> %.1f M positions/s, %.1f M bases/s (host buffers, PCIe included)" % (
            name, t * 1e3, (y - x + 1) / t / 1e6, len(seq) / t / 1e6))
    assert np.array_equal(p_out.array, out) and np.array_equal(p_skip.array, skip)
    # reads in, written records out: only the packed records (128 B per written position) cross PCIe on the way back
    ref2 = B.synth_ref_host(88172645463325252, x, nn + 2)
    p_ref2 = B.PinnedBuffer(nn + 2, np.uint8)
    p_ref2.array[:] = ref2
    p_rec = B.PinnedBuffer(nn, B.VCF_REC)
    for name, fn in (("block_records", lambda: c.block_records(tpl, seq, x, y, ref2, out=p_rec.array)),
                     ("block_records/pinned", lambda: c.block_records(p_tpl.array, p_seq.array, x, y, p_ref2.array, out=p_rec.array)),
                     ("  ... + statistics", lambda: c.block_records(p_tpl.array, p_seq.array, x, y, p_ref2.array, out=p_rec.array,
                                                                   with_stats=True))):
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            recs = fn()
            ts.append(time.perf_counter() - t0)
        t = min(ts)
        print("%-20s %.1f ms  -> %.1f M positions/s, %d records (%.1f %% of the positions, %.0f B/position back)" % (
            name, t * 1e3, nn / t / 1e6, len(recs), 100.0 * len(recs) / nn, 128.0 * len(recs) / nn))
