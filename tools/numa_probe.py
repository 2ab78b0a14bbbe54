#!/usr/bin/env python3
"""Where the GPU sits among the host's NUMA nodes, and what that does to pinned-buffer copies: H2D / D2H rates of
bsc_alloc_host buffers allocated (and first touched) from a thread bound to each node's CPUs in turn."""
import ctypes as C
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bs_call_amd as B

nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    cl = open(d + "/cpulist").read().strip()
    cpus = set()
    for part in cl.split(","):
        if part:
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
    nodes[int(d.rsplit("node", 1)[1])] = cpus
aff = os.sched_getaffinity(0)
print("nodes:", {k: len(v) for k, v in nodes.items()}, " affinity:", len(aff), "cpus on nodes", sorted({k for k, v in nodes.items() if v & aff}))
hip = C.CDLL("libamdhip64.so")
buf = C.create_string_buffer(64)
hip.hipDeviceGetPCIBusId(buf, 64, 0)
bdf = buf.value.decode().lower()
try:
    print("GPU", bdf, "numa_node", open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
except OSError as e:
    print("GPU", bdf, "numa_node unreadable:", e)
n = 512 << 20
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for node, cpus in nodes.items():
    use = cpus & aff
    if not use:
        print("node", node, ": no permitted cpu")
        continue
    os.sched_setaffinity(0, use)
    p = B.PinnedBuffer(n, np.uint8)
    p.array[:] = 1
    h = torch.from_numpy(p.array)
    for name, fn in (("D2H", lambda: h.copy_(d, non_blocking=True)), ("H2D", lambda: d.copy_(h, non_blocking=True))):
        best = 0
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t = time.perf_counter() - t0
            best = max(best, n / t / 1e9)
        print("buffer allocated from node %d cpus: %s %.1f GB/s" % (node, name, best), flush=True)
    del h, p
os.sched_setaffinity(0, aff)
