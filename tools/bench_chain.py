#!/usr/bin/env python3
"""HBM-resident print-side chain, pile-up -> VCF records -> site statistics, over one synthetic block:
  unfused  bsc_call_sites_device -> bsc_vcf_records_device -> bsc_vcf_stats_device   (gt_meth through HBM: 630 B / position)
  fused    bsc_chain_device (csrc/fused.hip)                                          (105 B in + 64 B out / position)
Prints one JSON line; run under `rocprofv3 --kernel-trace --stats` for the per-kernel times quoted in DESIGN.md.
usage: python tools/bench_chain.py [--sites N] [--coverage C] [--steps K] [--window W] [--no-unfused]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bs_call_amd as B

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=50_000_000)
ap.add_argument("--coverage", type=int, default=30)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--window", type=int, default=0, help="walk the block in windows of this many positions (0 = one call)")
ap.add_argument("--no-unfused", action="store_true")
ap.add_argument("--no-stats", action="store_true")
ap.add_argument("--warm", type=int, default=1, help="untimed launches in front (the first ~20 ms after an idle stretch run below the steady clock)")
args = ap.parse_args()
n, cov, x0 = args.sites, args.coverage, 1000
dev = torch.device("cuda:0")
stats = not args.no_stats


def timed(fn, steps):
    for _ in range(max(1, args.warm)):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


with B.SiteCaller() as c:
    d_cts = torch.empty((n + 2) * 104, dtype=torch.uint8, device=dev)
    d_ref = torch.empty(n + 2, dtype=torch.uint8, device=dev)
    d_vcf = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    c.synth_device(88172645463325254, x0, n + 2, cov, d_cts.data_ptr(), d_ref.data_ptr(), 0, None)
    torch.cuda.synchronize()
    res = {"positions": n, "coverage": cov, "with_stats": stats}
    if not args.no_unfused:
        d_out = torch.empty(n * 200, dtype=torch.uint8, device=dev)
        d_skip = torch.empty(n, dtype=torch.uint8, device=dev)

        def unfused():
            c.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out.data_ptr(), d_skip.data_ptr(), 200, None)
            c.vcf_records_device(d_out.data_ptr(), 200, d_skip.data_ptr(), d_ref.data_ptr(), n, x0, d_vcf.data_ptr())
            if stats:
                c.vcf_stats_device(d_vcf.data_ptr(), d_out.data_ptr(), 200, n)

        c.reset_site_stats()
        t = timed(unfused, args.steps)
        res["unfused_ms"] = t * 1e3
        res["unfused_Gpos_per_s"] = n / t / 1e9
        ref_core = d_vcf.clone()
        st_u = c.site_stats().copy()
        del d_out, d_skip

    wins = [(0, n)] if args.window <= 0 else [(s, min(args.window, n - s)) for s in range(0, n, args.window)]

    def fused():
        for first, m in wins:
            lc, lr = min(2, first), min(4, first)
            c.chain_device(d_cts.data_ptr() + (first - lc) * 104, d_ref.data_ptr() + (first - lr), x0, n, first, m,
                           d_vcf.data_ptr() + first * 64, with_stats=stats)

    c.reset_site_stats()
    c.set_profiling(True)
    t = timed(fused, args.steps)
    res["fused_ms"] = t * 1e3
    res["fused_Gpos_per_s"] = n / t / 1e9
    res["fused_device_ms_last_window"] = c.last_chain_ms()
    res["windows"] = len(wins)
    res["algorithmic_GBps"] = n * (105 + 64) / t / 1e9
    res["records"] = int(d_vcf.view(n, 64)[:, 4].sum())
    if not args.no_unfused:
        res["fused_equals_unfused_bytes"] = bool(torch.equal(ref_core, d_vcf))
        st_f = c.site_stats().copy()
        res["stats_snps_match"] = bool(int(st_f["snps"][0]) == int(st_u["snps"][0]))
    print(json.dumps(res))
