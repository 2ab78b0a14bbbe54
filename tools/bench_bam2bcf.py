#!/usr/bin/env python3
"""End-to-end host + device rate of integration/bam2bcf (BAM + FASTA -> BCF + report, plain C over the C ABI) on a synthetic
WGBS BAM (tools/make_bam.py wgbs_records: paired 2 x 100 bp, 30x).  Prints one JSON line.
usage: python tools/bench_bam2bcf.py [positions [insert]]   (insert, default 300: the template length — below 200 the mates overlap)"""
import importlib.util
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("make_bam", os.path.join(ROOT, "tools", "make_bam.py"))
W = importlib.util.module_from_spec(spec)
spec.loader.exec_module(W)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600_000
insert = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(5)
codes = rng.integers(1, 5, n).astype(np.uint8)
d = tempfile.mkdtemp(prefix="bam2bcf_")
t0 = time.time()
recs = W.wgbs_records(rng, codes, 0, n * 30 // 200, insert=insert)
bam, fa = os.path.join(d, "in.bam"), os.path.join(d, "ref.fa")
W.write_bam(bam, [("chrS", n)], recs)
with open(fa, "w") as f:
    f.write(">chrS\n")
    s = "".join("NACGT"[c] for c in codes)
    for o in range(0, n, 60):
        f.write(s[o : o + 60] + "\n")
gen_s = time.time() - t0
exe = os.path.join(ROOT, "bs_call_amd", "lib", "bam2bcf")
res = {}
digest = {}
import hashlib

# (inflate helper threads, mode): "device" = pre-processing AND BCF encoding on the device (bsc_block_bcf_raw, the default); "host_bcf" = the
# packed records come back and this thread encodes them (bsc_block_records_raw + bsc_bcf_block); "host_prep" = round 4's split
for threads, mode in ((0, "device"), (4, "device"), (4, "host_bcf"), (4, "host_prep")):
    best = None
    env = dict(os.environ, BAM2BCF_TIMING="1", BAM2BCF_THREADS=str(threads))
    if mode == "host_prep":
        env["BAM2BCF_HOST_PREP"] = "1"
    if mode == "host_bcf":
        env["BAM2BCF_HOST_BCF"] = "1"
    for _ in range(3):
        t0 = time.time()
        r = subprocess.run([exe, bam, fa, os.path.join(d, "out.bcf"), os.path.join(d, "rep.json")], capture_output=True, text=True, env=env)
        dt = time.time() - t0
        assert r.returncode == 0, r.stderr
        if best is None or dt < best:
            best, stages = dt, r.stderr.strip().splitlines()[-1]
    res[(threads, mode)] = (best, stages)
    digest[mode] = (hashlib.sha256(open(os.path.join(d, "out.bcf"), "rb").read()).hexdigest(), hashlib.sha256(open(os.path.join(d, "rep.json"), "rb").read()).hexdigest())
assert digest["device"] == digest["host_prep"] == digest["host_bcf"], "the three splits of the work wrote different files"
best = res[(4, "device")][0]
print(json.dumps({"no_inflate_threads": {"wall_s": round(res[(0, "device")][0], 3), "stages": res[(0, "device")][1]},
                  "bcf_encoding_on_the_host": {"wall_s_best_of_3": round(res[(4, "host_bcf")][0], 3), "stages": res[(4, "host_bcf")][1], "same_bcf_and_report_bytes": True},
                  "host_pre_processing_as_in_round_4": {"wall_s_best_of_3": round(res[(4, "host_prep")][0], 3), "stages": res[(4, "host_prep")][1], "same_bcf_and_report_bytes": True},
                  "positions": n, "insert": insert, "alignments": len(recs), "bam_bytes": os.path.getsize(bam), "bcf_bytes": os.path.getsize(os.path.join(d, "out.bcf")),
                  "generate_s": round(gen_s, 1), "bam2bcf_wall_s_best_of_3": round(best, 3), "positions_per_s": round(n / best),
                  "alignments_per_s": round(len(recs) / best), "stdout": r.stdout.strip(), "stages": res[(4, "device")][1],
                  "note": "whole process: context creation, BGZF inflate (4 helper threads) + pairing, GPU pre-processing + calling + BCF encoding (bsc_block_bcf_raw), the write, report; one host thread apart from the inflate helpers"}))
