#!/usr/bin/env python3
"""Device-resident chain pile-up -> gt_meth -> VCF records -> site statistics on HBM buffers; run under rocprofv3 --kernel-trace --stats to
get the per-kernel times quoted in DESIGN.md.  usage: python tools/bench_chain.py [sites] [coverage]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bs_call_amd as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
cov = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
with B.SiteCaller() as c:
    d_cts = torch.empty((n + 2) * 104, dtype=torch.uint8, device=dev)
    d_ref = torch.empty(n + 2, dtype=torch.uint8, device=dev)
    d_out = torch.empty(n * 200, dtype=torch.uint8, device=dev)
    d_skip = torch.empty(n, dtype=torch.uint8, device=dev)
    d_vcf = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    c.synth_device(88172645463325254, 1000, n + 2, cov, d_cts.data_ptr(), d_ref.data_ptr(), 0, None)
    torch.cuda.synchronize()
    for it in range(4):
        t0 = time.perf_counter()
        c.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out.data_ptr(), d_skip.data_ptr(), 200, None)
        c.vcf_records_device(d_out.data_ptr(), 200, d_skip.data_ptr(), d_ref.data_ptr(), n, 1000, d_vcf.data_ptr())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c.vcf_stats_device(d_vcf.data_ptr(), d_out.data_ptr(), 200, n)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
    emit = int(d_vcf.view(n, 64)[:, 4].sum())
    print("chain over %d positions: %.2f ms -> %.2f G positions/s; %d VCF records (%.1f %%)" % (n, dt * 1e3, n / dt / 1e9, emit, 100.0 * emit / n))
    st = c.site_stats()
    print("with site statistics: %.2f ms -> %.2f G positions/s; %d CpGs, %d records per launch" % (
        dt2 * 1e3, n / dt2 / 1e9, int(st["CpG_ref"][0] + st["CpG_nonref"][0]) // 4, int(st["snps"][0]) // 4))
