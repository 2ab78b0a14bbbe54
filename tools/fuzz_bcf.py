#!/usr/bin/env python3
"""Randomised soak of the BCF encoder on the device (csrc/bcfdev.hip) against the host encoder (csrc/bcf.c) and the independent Python
encoder (oracle/py_bcf.py): tests/test_gpu_bcf.py's generators with fresh seeds — random packed records of every size class, records that
are not written, names tables (short, long, with fillers), wide dictionary indices, block sizes around the 64-record tiles.
usage: python tools/fuzz_bcf.py [--minutes M] [--seed S]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bs_call_amd as B  # noqa: E402
from bs_call_amd import _lib  # noqa: E402
from oracle import py_bcf  # noqa: E402

import test_bcf as TB  # noqa: E402
import test_gpu_bcf as G  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=5.0)
ap.add_argument("--seed", type=int, default=900)
args = ap.parse_args()
t_end = time.time() + 60 * args.minutes
seed, ran, recs_done, two_pass = args.seed, 0, 0, 0
with B.SiteCaller() as c:
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        n = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, int(rng.integers(130, 3000))]))
        long_names = rng.random() < 0.3
        recs, names = G._random_block(rng, n, named=float(rng.choice([0.0, 0.2, 0.9])), long_names=long_names)
        ids = None
        if rng.random() < 0.4:
            ids = _lib.BcfIds(*[int(v) for v in rng.choice([0, 5, 127, 128, 300, 32767, 32768, 1_000_000, 2**31 - 1], 17)])
        rid = int(rng.integers(0, 3000))
        use_names = names if rng.random() < 0.8 else None
        want = G._host_stream(recs, rid, use_names, ids)
        got, total, bad = G._device_stream(c, recs, rid, use_names, ids)
        exp = b"".join(want)
        assert total == len(exp) and bad == 0, (seed, total, len(exp), bad)
        assert got[:total].tobytes() == exp, (seed, "device stream differs from the host encoder's")
        sizes = [len(w) for w in want]
        two_pass += int(any(sum(sizes[k : k + 64]) > 8192 for k in range(0, n, 64)))  # the wave image of the packed form (BCF_IMG_PACKED)
        if ids is None:  # the Python encoder knows the default dictionary only
            table = {} if use_names is None else {int(p): names[2][int(names[1][i]) : int(names[1][i + 1])][:63] for i, p in enumerate(names[0])}
            for j in range(0, n, max(1, n // 25)):
                r = recs[j]
                if r["core"]["emit"]:
                    rs = table.get(int(r["core"]["pos"]), b"") if r["rs_found"] else b""
                    assert want[j] == py_bcf.encode_record(TB._as_dict(r), rid, rs), (seed, j)
        ran += 1
        recs_done += n
        seed += 1
        if ran % 50 == 0:
            print("fuzz_bcf: %d blocks, %d records, %d with a tile in several parts, seed %d" % (ran, recs_done, two_pass, seed), flush=True)
print("fuzz_bcf: %d blocks (%d records, %d blocks with a tile in several parts) equal to the host and the Python encoder; seeds %d .. %d" % (ran, recs_done, two_pass, args.seed, seed - 1))
