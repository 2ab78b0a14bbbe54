#!/usr/bin/env python3
"""Host-buffer throughput of bsc_call_sites (PCIe included): pageable vs pinned buffers.  Never bench.py's `value`.
usage: python tools/bench_host.py [sites]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bs_call_amd as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
with B.SiteCaller() as c:
    dev = torch.device("cuda:0")
    d_cts = torch.empty(n * 104, dtype=torch.uint8, device=dev)
    d_ref = torch.empty(n, dtype=torch.uint8, device=dev)
    c.synth_device(88172645463325254, 0, n, 30, d_cts.data_ptr(), d_ref.data_ptr(), 0, None)
    torch.cuda.synchronize()
    pile = d_cts.cpu().numpy().view(B.PILEUP)
    ref = d_ref.cpu().numpy()
    del d_cts, d_ref
    # pageable
    out = np.zeros(n, dtype=B.GT_METH)
    skip = np.zeros(n, dtype=np.uint8)
    c.call_sites(pile, ref, out=out, skip=skip)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); c.call_sites(pile, ref, out=out, skip=skip); ts.append(time.perf_counter() - t0)
    print("pageable buffers: %.1f ms -> %.1f M positions/s (%.1f GB/s over the bus)" % (min(ts) * 1e3, n / min(ts) / 1e6, n * 306 / min(ts) / 1e9))
    ref_out = out.copy()
    # pinned
    pb = [B.PinnedBuffer(n, B.PILEUP), B.PinnedBuffer(n, np.uint8), B.PinnedBuffer(n, B.GT_METH), B.PinnedBuffer(n, np.uint8)]
    pb[0].array[:] = pile
    pb[1].array[:] = ref
    c.call_sites(pb[0].array, pb[1].array, out=pb[2].array, skip=pb[3].array)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); c.call_sites(pb[0].array, pb[1].array, out=pb[2].array, skip=pb[3].array); ts.append(time.perf_counter() - t0)
    print("pinned buffers:   %.1f ms -> %.1f M positions/s (%.1f GB/s over the bus)" % (min(ts) * 1e3, n / min(ts) / 1e6, n * 306 / min(ts) / 1e9))
    assert pb[2].array.tobytes() == ref_out.tobytes()
    for b in pb:
        b.free()
