#!/usr/bin/env python3
"""Randomised soak of the reads -> pile-up -> gt_meth -> record path on the GPU: random model parameters x random blocks
(tests/test_gpu_accumulate.py::test_fuzz_parameters_and_blocks with fresh seeds) and adversarial template lists
(::test_adversarial_template_lists), each against the CPU oracle.  usage: python tools/fuzz_block.py [--minutes M] [--seed S]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bs_call_amd as B
from oracle import loader as O
from tests import test_gpu_accumulate as TA

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=5.0)
ap.add_argument("--seed", type=int, default=100)
args = ap.parse_args()
exact = O.libm_exact()
t_end = time.time() + 60 * args.minutes
seed = args.seed
with B.SiteCaller() as c:
    while time.time() < t_end:
        try:
            TA.test_fuzz_parameters_and_blocks.__wrapped__(O, seed, exact) if hasattr(TA.test_fuzz_parameters_and_blocks, "__wrapped__") else TA.test_fuzz_parameters_and_blocks(O, seed, exact)
            TA.test_adversarial_template_lists.__wrapped__(c, O, seed) if hasattr(TA.test_adversarial_template_lists, "__wrapped__") else TA.test_adversarial_template_lists(c, O, seed)
        except AssertionError as e:
            print("MISMATCH at seed %d: %s" % (seed, str(e)[:300]), flush=True)
            sys.exit(1)
        seed += 1
        if (seed - args.seed) % 20 == 0:
            print("seed %d ok" % seed, flush=True)
print("fuzz done: seeds %d .. %d, no difference" % (args.seed, seed - 1))
