#!/usr/bin/env python3
"""Randomised soak of the fused chain on the GPU: random blocks (WGBS-like from the generator, or adversarial class counts
with ties / deep sites / N runs / dbSNP flags), random window cuts, random printing parameters — fused records and
statistics against the three unfused kernels, and (every few rounds) both against the CPU oracle.  Prints a progress line
per round; stops at the first difference with the seed that reproduces it.
usage: python tools/fuzz_chain.py [--minutes M] [--seed S]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bs_call_amd as B
from bs_call_amd.abi import SITE_STATS
from oracle import loader as O
from tests import test_gpu_chain as T

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=5.0)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
tables = O.Tables()
flav = O.LIBM if O.libm_exact() else O.BSM
t_end = time.time() + 60 * args.minutes
rnd = 0
with B.SiteCaller() as c:
    while time.time() < t_end:
        seed = args.seed * 1_000_003 + rnd
        try:
            n, nw = T.fuzz_round(c, O, tables, flav, seed, rnd)
        except AssertionError as e:
            print("MISMATCH seed=%d round=%d: %s" % (seed, rnd, str(e)[:600]), flush=True)
            sys.exit(1)
        rnd += 1
        if rnd % 25 == 0:
            print("round %d ok (n=%d, %d windows)" % (rnd, n, nw), flush=True)
print("fuzz done: %d rounds, no difference" % rnd)
