#!/usr/bin/env python3
"""The BCF encoder on the device (csrc/bcfdev.hip) at size: packed records of a synthetic contig (L-pileup generator -> calling kernel ->
record formation -> packing, all in HBM) -> bsc_bcf_block_device, timed with events on the launch stream; the host encoder
(bsc_bcf_block, one thread) on a sample of the same records beside it.  Prints one JSON line.
usage: python tools/bench_bcf.py [--sites N] [--coverage C] [--steps K]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bs_call_amd as B  # noqa: E402
from bs_call_amd import _lib, vcf  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=20_000_000)
ap.add_argument("--coverage", type=int, default=30)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--host-sample", type=int, default=400_000)
a = ap.parse_args()
n, cov = a.sites, a.coverage
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
with B.SiteCaller() as c:
    d_cts = torch.empty((n + 2) * 104, dtype=torch.uint8, device=dev)
    d_ref = torch.empty(n + 2, dtype=torch.uint8, device=dev)
    d_out = torch.empty(n * 200, dtype=torch.uint8, device=dev)
    d_skip = torch.empty(n, dtype=torch.uint8, device=dev)
    d_vcf = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    c.synth_device(88172645463325252, 10_000, n + 2, cov, d_cts.data_ptr(), d_ref.data_ptr(), 0, st)
    c.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out.data_ptr(), d_skip.data_ptr(), 200, st)
    c.vcf_records_device(d_out.data_ptr(), 200, d_skip.data_ptr(), d_ref.data_ptr(), n, 10_000, d_vcf.data_ptr(), stream=st)
    del d_cts
    d_rec = torch.empty(n * 128, dtype=torch.uint8, device=dev)
    c.vcf_compact_device(d_vcf.data_ptr(), d_out.data_ptr(), 200, n, d_rec.data_ptr(), n, d_cnt.data_ptr(), stream=st)
    n_rec = int(d_cnt.item())
    del d_out, d_vcf, d_skip
    cap = n_rec * 160 + 4096
    d_bcf = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_tot = torch.zeros(3, dtype=torch.int64, device=dev)
    for _ in range(3):
        c.bcf_block_device(d_rec.data_ptr(), d_cnt.data_ptr(), n, 0, d_bcf.data_ptr(), cap, d_tot.data_ptr(), stream=st)
    torch.cuda.synchronize()
    nbytes = int(d_tot[0].item())
    assert nbytes <= cap and int(d_tot[1].item()) == 0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    ev[0].record()
    for k in range(a.steps):
        c.bcf_block_device(d_rec.data_ptr(), d_cnt.data_ptr(), n, 0, d_bcf.data_ptr(), cap, d_tot.data_ptr(), stream=st)
        ev[k + 1].record()
    torch.cuda.synchronize()
    ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(a.steps)]
    # the host encoder on a sample, and the bytes compared
    m = min(a.host_sample, n_rec)
    recs = d_rec[: m * 128].cpu().numpy().view(B.VCF_REC)
    t0 = time.time()
    want = vcf.bcf_block(recs, 0)
    host_s = time.time() - t0
    got = d_bcf[: len(want)].cpu().numpy().tobytes()
    same = got == want
alg = n_rec * 128 * 2 + nbytes  # the records read by both kernels, the stream written once
print(json.dumps({"sites": n, "coverage": cov, "records": n_rec, "bcf_bytes": nbytes, "bytes_per_record": round(nbytes / max(n_rec, 1), 2),
                  "device_ms_avg": round(float(np.mean(ms)), 4), "device_ms_min": round(float(np.min(ms)), 4),
                  "records_per_s": round(n_rec / (np.mean(ms) * 1e-3)), "algorithmic_bytes": alg,
                  "achieved_GBps": round(alg / (np.mean(ms) * 1e-3) / 1e9, 1), "frac_of_8TBps": round(alg / (np.mean(ms) * 1e-3) / 8e12, 4),
                  "host_encoder": {"records": m, "seconds": round(host_s, 4), "records_per_s": round(m / host_s), "note": "bsc_bcf_block, one host thread"},
                  "device_over_host": round((n_rec / (np.mean(ms) * 1e-3)) / (m / host_s), 1), "first_records_equal_host": bool(same),
                  "note": "size kernel (reads the records) + scan of the tile sums + write kernel (reads the records, writes the stream); launches of "
                          "bsc_bcf_block_device back to back, events on the launch stream"}))
assert same
