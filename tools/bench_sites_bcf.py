#!/usr/bin/env python3
"""The BCF encoder in the form the block entries run it — over the reads-in chain's per-position arrays, the stream sized from the chain's
length bytes (bsc_reads_chain_len_device, then bsc_bcf_sites_len_device) — alone, device-resident, at config size: for the kernel trace
and the PMC passes of tools/prof_r06.sh (the only launches of bsc_bcf_size_bytes_kernel / bsc_bcf_write_kernel in the process are this
form's).  Prints one JSON line.  usage: python tools/bench_sites_bcf.py [--sites N] [--coverage C] [--steps K]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bs_call_amd as B
from bs_call_amd import reads as R

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=50_000_000)
ap.add_argument("--coverage", type=int, default=30)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
x = 1000
tpl, seq, y = R.synth_block(88172645463325252 + 2, x, a.sites, a.coverage)
ref = B.synth_ref_host(88172645463325252 + 2, x, y - x + 3)
n = y - x + 1
st = torch.cuda.current_stream().cuda_stream
with B.SiteCaller() as c:
    up = lambda v: torch.from_numpy(v.view(np.uint8).reshape(-1).copy()).to(dev)
    d_tpl, d_seq, d_ref = up(tpl), up(seq), up(ref)
    d_core = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    d_aux = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    c.reads_chain_len_device(d_tpl.data_ptr(), len(tpl), d_seq.data_ptr(), seq.size, x, y, d_ref.data_ptr(), d_core.data_ptr(), d_aux.data_ptr(), d_len.data_ptr(),
                             stream=st)
    c.block_status(st)
    cap = n * 96 + 4096
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_tot = torch.zeros(3, dtype=torch.int64, device=dev)
    for _ in range(2):
        c.bcf_sites_len_device(d_core.data_ptr(), d_aux.data_ptr(), d_len.data_ptr(), n, 0, d_out.data_ptr(), cap, d_tot.data_ptr(), stream=st)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * a.steps)]
    for k in range(a.steps):
        ev[2 * k].record()
        c.bcf_sites_len_device(d_core.data_ptr(), d_aux.data_ptr(), d_len.data_ptr(), n, 0, d_out.data_ptr(), cap, d_tot.data_ptr(), stream=st)
        ev[2 * k + 1].record()
    torch.cuda.synchronize()
    ms = [ev[2 * k].elapsed_time(ev[2 * k + 1]) for k in range(a.steps)]
    tot = d_tot.cpu().numpy()
    nb = int(tot[0])
    # a checksum of the stream, on the device (position-weighted, so that moved bytes show): for A/B runs of two forms of the encoder
    w = d_out[: nb - nb % 8].view(torch.int64)
    out_sum = int((w * (torch.arange(w.numel(), device=dev, dtype=torch.int64) | 1)).sum().item()) if nb >= 8 else 0
print(json.dumps({"out_sum": out_sum, "sites": n, "coverage": a.coverage, "records": int(tot[2]), "bcf_bytes": int(tot[0]), "refused": int(tot[1]), "stage_ms_avg": float(np.mean(ms)),
                  "stage_ms_min": float(np.min(ms)), "algorithmic_bytes": int(n + n + 128 * int(tot[2]) + int(tot[0]))}))
