#!/usr/bin/env python3
"""The passes behind the chain kernel as the block entries run them (bsc_block_records: packing; bsc_block_bcf: BCF encoding), to be run
under `rocprofv3 --kernel-trace --stats` (tools/r05_tail.sh): one block of synthetic reads through both entries, three times each.
BSC_NO_EMIT_BYTES in the environment: without the chain's emit bytes (the A/B).  usage: python tools/bench_tail.py [--sites N]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bs_call_amd as B  # noqa: E402
from bs_call_amd.reads import synth_block  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=20_000_000)
ap.add_argument("--coverage", type=int, default=30)
a = ap.parse_args()
x = 1000
tpl, seq, y = synth_block(88172645463325252 + 2, x, a.sites, a.coverage)
ref = B.synth_ref_host(88172645463325252 + 2, x, y - x + 3)
with B.SiteCaller() as c:
    for _ in range(3):
        recs = c.block_records(tpl, seq, x, y, ref)
    for _ in range(3):
        blob, n_rec = c.block_bcf(tpl, seq, x, y, ref, 0)
print("%d positions, %d records, %d BCF bytes, emit bytes %s" % (y - x + 1, len(recs), len(blob), "off" if os.environ.get("BSC_NO_EMIT_BYTES") else "on"))
assert n_rec == len(recs)
