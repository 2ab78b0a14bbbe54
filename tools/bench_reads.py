#!/usr/bin/env python3
"""HOT LOOP A (reads -> pile-up, reference src/call_genotypes.c:180-226) on device-resident L-reads at config sizes:
bsc_accumulate_device timed with HIP events on its stream (bsc_last_accumulate_ms) — template checks + read descriptors,
ordering, tile search, accumulate — and, when the library has it, the reads-in / records-out chain
(bsc_reads_chain_device).  Prints one JSON line; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
usage: python tools/bench_reads.py [--sites N] [--coverage C] [--steps K] [--no-chain] [--no-check]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bs_call_amd as B
from bs_call_amd import reads as R

SEED = 88172645463325252 + 2
ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=50_000_000)
ap.add_argument("--coverage", type=int, default=30)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--chunk", type=int, default=1_000_000)
ap.add_argument("--no-chain", action="store_true")
ap.add_argument("--no-check", action="store_true")
ap.add_argument("--one-kernel", action="store_true", help="the reads-in chain kernel (bsc_set_reads_fused) instead of the default two-kernel form")
ap.add_argument("--warm", type=int, default=2, help="untimed launches in front of each timed loop (the first ~20 ms after an idle stretch run below the steady clock)")
args = ap.parse_args()
dev = torch.device("cuda:0")
x = 1000
t0 = time.time()
tpl, seq, y = R.synth_block(SEED, x, args.sites, args.coverage, chunk=args.chunk)
gen_s = time.time() - t0
n = y - x + 1
n_pad = (n + 63) // 64 * 64
res = {"positions": n, "coverage": args.coverage, "templates": int(len(tpl)), "bases": int(seq.size), "generate_s": gen_s}
with B.SiteCaller() as c:
    d_tpl = torch.from_numpy(tpl.view(np.uint8).reshape(-1)).to(dev)
    d_seq = torch.from_numpy(seq).to(dev)
    d_cts = torch.empty(n_pad * 104, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    c.set_profiling(True)
    c.set_reads_fused(args.one_kernel)
    res["reads_path"] = "one kernel (reads-in chain)" if args.one_kernel else "two kernels (site summaries through HBM)"
    ms, wall = [], []
    for it in range(args.warm + args.steps):
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        c.accumulate_device(d_tpl.data_ptr(), len(tpl), d_seq.data_ptr(), seq.size, x, y, d_cts.data_ptr(), stream)
        c.block_status(stream)
        if it >= args.warm:
            wall.append(time.perf_counter() - w0)
            ms.append(c.last_accumulate_ms())
    k = float(np.mean(ms))
    bytes_in = R.algorithmic_bytes_in(tpl, seq)
    res["accumulate"] = {
        "device_ms_avg": k,
        "device_ms_min": float(np.min(ms)),
        "wall_ms_median": float(np.median(wall)) * 1e3,
        "G_positions_per_s": n / (k * 1e-3) / 1e9,
        "G_bases_per_s": seq.size / (k * 1e-3) / 1e9,
        "algorithmic_bytes": {"in": bytes_in, "out": n * 104, "per_position": (bytes_in + n * 104) / n},
        "algorithmic_GBps": (bytes_in + n * 104) / (k * 1e-3) / 1e9,
    }
    if not args.no_check:
        # positions well inside the first chunk are covered by that chunk's templates only: compare with the CPU oracle
        from oracle import loader as O

        m = min(args.chunk, args.sites)
        t1, s1, y1 = R.synth_block(SEED, x, m, args.coverage, chunk=args.chunk)
        _rc, exp = O.accumulate(t1, s1, x, y1, 20)
        keep = (m - 400) if args.sites > m else (y1 - x + 1)
        got = d_cts[: keep * 104].cpu().numpy().view(B.PILEUP)
        res["accumulate"]["first_chunk_equals_oracle"] = bool(got.tobytes() == exp[:keep].tobytes())
        res["accumulate"]["sum_n"] = int(d_cts.view(torch.int32).view(n_pad, 26)[:n, 16].sum(dtype=torch.int64))
    if not args.no_chain and hasattr(c, "reads_chain_device"):
        ref = B.synth_ref_host(SEED, x, n + 2)
        d_ref = torch.from_numpy(ref).to(dev)
        d_core = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        ms, wall = [], []
        for it in range(args.warm + args.steps):
            torch.cuda.synchronize()
            w0 = time.perf_counter()
            c.reads_chain_device(d_tpl.data_ptr(), len(tpl), d_seq.data_ptr(), seq.size, x, y, d_ref.data_ptr(), d_core.data_ptr(),
                                 with_stats=True, stream=stream)
            c.block_status(stream)
            if it >= args.warm:
                wall.append(time.perf_counter() - w0)
                ms.append(c.last_reads_chain_ms())
        k = float(np.mean(ms))
        res["reads_chain"] = {
            "device_ms_avg": k,
            "device_ms_min": float(np.min(ms)),
            "wall_ms_median": float(np.median(wall)) * 1e3,
            "G_positions_per_s": n / (k * 1e-3) / 1e9,
            "algorithmic_bytes": {"in": bytes_in + n, "out": n * 64, "per_position": (bytes_in + n + n * 64) / n},
            "algorithmic_GBps": (bytes_in + n + n * 64) / (k * 1e-3) / 1e9,
            "records": int(d_core.view(n, 64)[:, 4].sum()),
        }
        if not args.no_check:
            # the unfused route over the same block: accumulate -> fused chain on the pile-up
            d_core2 = torch.empty(n * 64, dtype=torch.uint8, device=dev)
            d_cts2 = torch.zeros((n_pad + 2) * 104, dtype=torch.uint8, device=dev)
            c.accumulate_device(d_tpl.data_ptr(), len(tpl), d_seq.data_ptr(), seq.size, x, y, d_cts2.data_ptr(), stream)
            c.chain_device(d_cts2.data_ptr(), d_ref.data_ptr(), x, n, 0, n, d_core2.data_ptr(), with_stats=False, stream=stream)
            torch.cuda.synchronize()
            res["reads_chain"]["equals_accumulate_then_chain_bytes"] = bool(torch.equal(d_core, d_core2))
print(json.dumps(res))
