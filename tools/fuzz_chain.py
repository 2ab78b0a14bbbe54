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
        rng = np.random.default_rng(seed)
        n = int(rng.choice([1, 2, 59, 60, 61, 119, 500, 5_000, 60_000, 200_000]))
        x0 = int(rng.choice([1, 2, 3, 1000, 4_000_000_000 - 300_000]))
        kind = rng.random()
        if kind < 0.5:
            pile, ref2 = B.synth_pileup_host(seed, x0 + 7, n + 2, int(rng.choice([1, 5, 30, 200, 600])), int(rng.integers(0, 2)))
            pile = pile[:n]
        else:
            pile = np.zeros(n, dtype=B.PILEUP)
            scale = rng.choice([1, 1, 3, 10, 40, 400], size=n)
            mask = rng.random((n, 2, 8)) < rng.choice([0.1, 0.3, 0.6, 1.0], size=(n, 1, 1))
            cnt = (rng.integers(0, 8, size=(n, 2, 8)) * scale[:, None, None] * mask).astype(np.uint32)
            cnt[rng.random(n) < rng.choice([0.0, 0.05, 0.5])] = 0
            pile["counts"] = cnt
            tot = cnt.sum(axis=1)
            pile["n"] = tot.sum(axis=1)
            pile["quality"] = np.minimum((tot * rng.integers(20, 44, size=(n, 8))).astype(np.float32), 43.0 * tot)
            pile["mapq2"] = (pile["n"] * rng.choice([0, 1, 400, 1521, 3600], size=n)).astype(np.float32)
            ref2 = rng.integers(0 if rng.random() < 0.3 else 1, 5, size=n + 2).astype(np.uint8)
        flags = rng.choice([0, 0, 0, 1, 3], size=n).astype(np.uint8) if rng.random() < 0.5 else None
        allp = bool(rng.integers(0, 2)) if rng.random() < 0.3 else False
        reg = (1, 0xFFFFFFFF) if rng.random() < 0.7 else (x0 + n // 4, x0 + 3 * n // 4)
        cuts = sorted(set(int(v) for v in rng.integers(0, n + 1, int(rng.integers(0, 6)))) | {0, n})
        wins = [(a, b - a) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        exp, est, ecnt = T._unfused(c, pile, ref2, x0, dbsnp=flags, all_positions=allp, reg=reg)
        got, gst, gcnt = T._fused(c, pile, ref2, x0, wins, dbsnp=flags, all_positions=allp, reg=reg)
        try:
            T._same_core(got, exp, "fused vs unfused")
            T._same_stats(gst, est)
            assert gcnt == ecnt
            if rnd % 4 == 0 and n <= 60_000:
                gtm, skip = O.call_sites(pile, ref2[:n], tables, flav, -4)
                st = np.zeros(1, dtype=SITE_STATS)
                ocore = O.vcf_block_stats(gtm, skip, ref2, x0, st, np.zeros(2, dtype=np.uint32), tables.lfact_store, allp, reg[0], reg[1], flags)
                T._same_core(got, ocore, "fused vs oracle")
                T._same_stats(gst, st[0])
        except AssertionError as e:
            print("MISMATCH seed=%d round=%d n=%d x0=%d kind=%.2f wins=%s: %s" % (seed, rnd, n, x0, kind, wins, str(e)[:400]), flush=True)
            sys.exit(1)
        rnd += 1
        if rnd % 25 == 0:
            print("round %d ok (n=%d, %d windows)" % (rnd, n, len(wins)), flush=True)
print("fuzz done: %d rounds, no difference" % rnd)
