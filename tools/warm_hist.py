#!/usr/bin/env python3
"""Per-launch device times of the calling kernel after a short and a long warm-up: the first ~30 ms of launches after an idle
period run below the steady clock.  usage: python tools/warm_hist.py [sites] [coverage]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bs_call_amd as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
cov = int(sys.argv[2]) if len(sys.argv) > 2 else 200
d_cts = torch.empty((n + 2) * 104, dtype=torch.uint8, device="cuda")
d_ref = torch.empty(n + 2, dtype=torch.uint8, device="cuda")
d_out = torch.empty(n * 200, dtype=torch.uint8, device="cuda")
d_skip = torch.empty(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for warm in (3, 5, 40, 5):
    with B.SiteCaller() as c:
        c.synth_device(1, 0, n + 2, cov, d_cts.data_ptr(), d_ref.data_ptr(), 0, s)
        torch.cuda.synchronize()
        c.set_profiling(True)
        for _ in range(warm + 20):
            c.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out.data_ptr(), d_skip.data_ptr(), 200, s)
        torch.cuda.synchronize()
        h = [x[0] for x in c.kernel_ms_history(20)]
        print("%d positions at %dx, %2d warm-up launches: %s  mean %.4f ms" % (n, cov, warm, " ".join("%.3f" % x for x in h), np.mean(h)), flush=True)
