#!/usr/bin/env python3
"""Randomised soak of the reads -> records paths on the GPU against the CPU oracle chain (accumulate -> call -> print_vcf
restatement): random blocks (1 .. 60 000 positions, 5x .. 400x, odd sizes around the tile geometry, blocks without reads),
random printing parameters and dbSNP flags, through
  * bsc_block_records in its two-kernel form (accumulate kernel's summary form -> chain kernel's summary-in form),
  * bsc_block_records in its one-kernel form (reads-in chain; bsc_set_reads_fused),
  * bsc_blocks_records (several blocks in one launch sequence) — against the two above block by block, the statistics included —
    and bsc_blocks_bcf_submit / _fetch (the same launch sequence, one BCF stream back) against the host encoder over those records.
Prints a progress line per round; stops at the first difference with the seed that reproduces it.
usage: python tools/fuzz_reads.py [--minutes M] [--seed S]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bs_call_amd as B
from bs_call_amd import vcf
from oracle import loader as O
from tests import test_gpu_blocks as T

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=5.0)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
O.build()
tables, exact = O.Tables(), O.libm_exact()
assert exact, "host libm differs from the replica: nothing to compare bytes with"
t_end = time.time() + 60 * args.minutes
rng = np.random.default_rng(args.seed)
c2, c1 = B.SiteCaller(), B.SiteCaller()
c1.set_reads_fused(True)
rounds = blocks_done = positions = 0
SIZES = [1, 2, 59, 60, 61, 62, 63, 64, 65, 121, 122, 123, 124, 125, 126, 127, 128, 129]
while time.time() < t_end:
    seed = int(rng.integers(1, 2**31))
    r = np.random.default_rng(seed)
    nb = int(r.integers(1, 24))
    blocks, refs, dbs, pos = [], [], [], int(r.integers(1, 5000))
    for i in range(nb):
        n = int(r.choice(SIZES)) if r.random() < 0.3 else int(r.integers(1, 60_000 if r.random() < 0.1 else 6_000))
        cov = int(r.choice([5, 10, 30, 30, 60, 400])) if n < 3000 else int(r.choice([5, 10, 30]))
        tpl, seq, x, y = T._block(seed + i, pos + int(r.integers(3, 300)), n, cov)
        if r.random() < 0.05:
            tpl = tpl[:0]
        blocks.append((tpl, seq, x, y))
        refs.append(B.synth_ref_host(seed + i, x, y - x + 3))
        dbs.append(r.choice([0, 1, 3], size=y - x + 1, p=[0.9, 0.05, 0.05]).astype(np.uint8))
        pos = y
    use_db = r.random() < 0.5
    kw = [dict(), dict(all_positions=True), dict(reg_start=blocks[0][2] + 10, reg_stop=blocks[-1][3] - 10)][int(r.integers(0, 3))]
    db = dbs if use_db else None
    for c in (c1, c2):
        c.reset_site_stats()
    one = [c1.block_records(t, s, x, y, refs[i], dbsnp=None if db is None else db[i], with_stats=True, **kw).copy() for i, (t, s, x, y) in enumerate(blocks)]
    two = [c2.block_records(t, s, x, y, refs[i], dbsnp=None if db is None else db[i], with_stats=True, **kw).copy() for i, (t, s, x, y) in enumerate(blocks)]
    st1, st2 = c1.site_stats().copy(), c2.site_stats().copy()
    c2.reset_site_stats()
    got, per = c2.blocks_records(blocks, refs, dbsnp=db, with_stats=True, **kw)
    st3 = c2.site_stats().copy()
    ok = [int(v) for v in per] == [len(v) for v in one] and got.tobytes() == np.concatenate(one).tobytes() and np.concatenate(two).tobytes() == got.tobytes()
    for f in B.SITE_STATS.names:
        if f.endswith("_meth"):
            ok = ok and np.allclose(st1[f], st2[f], rtol=1e-12, atol=0) and np.allclose(st1[f], st3[f], rtol=1e-12, atol=0)
        else:
            ok = ok and st1[f].tobytes() == st2[f].tobytes() == st3[f].tobytes()
    if ok:  # ... and the oracle chain, a block of the round (all of them every eighth round)
        for i in ([int(r.integers(0, nb))] if rounds % 8 else range(nb)):
            t, s, x, y = blocks[i]
            rc, pile = O.accumulate(t, s, x, y, 20)
            gtm, skip = O.call_sites(pile, refs[i][: y - x + 1], tables, O.LIBM, 1)
            core = O.vcf_block(gtm, skip, refs[i], x, dbsnp=None if db is None else db[i], **kw)
            sel = core["emit"] == 1
            ok = ok and rc == 0 and one[i]["core"].tobytes() == core[sel].tobytes() and (one[i]["counts"] == gtm["counts"][sel]).all() \
                and (one[i]["mq"] == gtm["mq"][sel]).all() and (one[i]["qual"] == gtm["qual"][sel]).all()
    if ok:  # ... and the bytes form of the batch (bsc_blocks_bcf_*): ONE stream = the host encoder (csrc/bcf.c) over the records above, in order
        stream, n_rec = c2.blocks_bcf(blocks, refs, 7, dbsnp=db, **kw)
        ok = n_rec == len(got) and stream == vcf.bcf_block(got, 7)
    if not ok:
        print("DIFFERENCE at seed %d (round %d): %d blocks, %s, dbsnp=%s" % (seed, rounds, nb, kw, use_db), flush=True)
        sys.exit(1)
    rounds += 1
    blocks_done += nb
    positions += sum(y - x + 1 for _, _, x, y in blocks)
    if rounds % 20 == 0:
        print("%d rounds, %d blocks, %.1f M positions: one kernel = two kernels = batched = oracle" % (rounds, blocks_done, positions / 1e6), flush=True)
print("done: %d rounds, %d blocks, %.1f M positions, no difference (seed %d)" % (rounds, blocks_done, positions / 1e6, args.seed))
