#!/usr/bin/env python3
"""The launch-bound end of the path (BASELINE configs[0]: 1 Mb at 10x — a step is two short kernels): bsc_call_sites_device queued call by
call against the same calls captured once in a HIP graph (torch.cuda.CUDAGraph around the library call: every launch of the entry goes to the
stream it is handed, its workspaces are grow-only and stand after the first call) and replayed.  Same bytes; one JSON line.
usage: python tools/bench_graph.py [--sites N] [--coverage C] [--steps K] [--per-graph G]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bs_call_amd as B  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=1_000_000)
ap.add_argument("--coverage", type=int, default=10)
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--per-graph", type=int, default=20, help="calls captured in one graph")
a = ap.parse_args()
n, dev = a.sites, torch.device("cuda:0")
with B.SiteCaller() as c:
    side = torch.cuda.Stream()
    d_cts = torch.empty((n + 2) * 104, dtype=torch.uint8, device=dev)
    d_ref = torch.empty(n + 2, dtype=torch.uint8, device=dev)
    d_out = torch.empty(n * 200, dtype=torch.uint8, device=dev)
    d_skip = torch.empty(n, dtype=torch.uint8, device=dev)
    d_out2 = torch.empty(n * 200, dtype=torch.uint8, device=dev)
    d_skip2 = torch.empty(n, dtype=torch.uint8, device=dev)
    with torch.cuda.stream(side):
        st = side.cuda_stream
        c.synth_device(88172645463325252, 0, n + 2, a.coverage, d_cts.data_ptr(), d_ref.data_ptr(), 0, st)
        for _ in range(20):  # warm-up: workspaces allocated, clocks up
            c.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out.data_ptr(), d_skip.data_ptr(), 200, st)
        side.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            c.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out.data_ptr(), d_skip.data_ptr(), 200, st)
        side.synchronize()
        plain_s = (time.perf_counter() - t0) / a.steps
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(a.per_graph):
                c.call_sites_device(d_cts.data_ptr(), d_ref.data_ptr(), n, d_out2.data_ptr(), d_skip2.data_ptr(), 200, side.cuda_stream)
        for _ in range(3):
            g.replay()
        side.synchronize()
        reps = max(1, a.steps // a.per_graph)
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        side.synchronize()
        graph_s = (time.perf_counter() - t0) / (reps * a.per_graph)
    same = bool(torch.equal(d_out, d_out2) and torch.equal(d_skip, d_skip2))
print(json.dumps({"workload": "%d positions at %dx, pile-ups resident in HBM (BASELINE configs[0] shape)" % (n, a.coverage), "steps": a.steps,
                  "call_by_call": {"us_per_step": round(plain_s * 1e6, 2), "positions_per_s": round(n / plain_s)},
                  "hip_graph": {"calls_per_graph": a.per_graph, "us_per_step": round(graph_s * 1e6, 2), "positions_per_s": round(n / graph_s)},
                  "speedup": round(plain_s / graph_s, 3), "same_bytes": same}))
assert same
