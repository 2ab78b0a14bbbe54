#!/usr/bin/env python3
"""Randomised soak of the DEVICE BAM reader (csrc/bamstream.c + csrc/bamdev.hip) on the GPU box: random BAM files — the adversarial
generator of tests/test_bam.py (re-used names, odd pairs, negative mate positions: the one-lane replay's cases) and the ordinary one of
tests/test_bamdev_emul.py (sorted, paired: the parallel kernels' cases), random BGZF block sizes, reader parameters, slab / pass sizes, the
replay forced at random, contig selections — against the host reader csrc/bamio.c (every file) and the Python restatement oracle/py_bam.py
(every fourth file): blocks, templates, read bytes, mismatch lists and filter counters must be equal, or both must refuse the file.
usage: python tools/fuzz_bamdev.py [--minutes M] [--seed S]"""
import argparse
import importlib.util
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import bs_call_amd as B  # noqa: E402
from bs_call_amd.bamdev import DeviceBamReader  # noqa: E402
from bs_call_amd.caller import BscError  # noqa: E402


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


TB = _load("test_bam_fz", "tests/test_bam.py")
TE = _load("test_bamdev_emul_fz", "tests/test_bamdev_emul.py")
W = TB.W
ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=2.0)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t_end = time.time() + 60 * a.minutes
n_files = n_refused = n_py = n_blocks = n_tpl = n_replay = 0
ENV = ("BSC_BAMDEV_REPLAY", "BSC_BAMDEV_SLAB_KB", "BSC_BAMDEV_PASS_KB")
with B.SiteCaller() as caller, tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "f.bam")
    while time.time() < t_end:
        sane = rng.random() < 0.5
        n = int(rng.integers(1, 3000 if sane else 600))
        recs = TE._sane_records(rng, n, dup_rate=float(rng.choice([0.0, 0.05, 0.3, 0.7]))) if sane else TB._random_records(rng, n)
        W.write_bam(path, TB.REFS, recs, block=int(rng.choice([0xFF00, 333, 777, 4096, 20000])))
        kw = {}
        if rng.random() < 0.3:
            kw["keep_unmatched"] = True
        if rng.random() < 0.3:
            kw["keep_duplicates"] = True
        elif rng.random() < 0.2:
            kw["ignore_duplicates"] = True
        if rng.random() < 0.3:
            kw["mapq_thresh"] = int(rng.choice([0, 10, 30]))
        if rng.random() < 0.3:
            kw["max_template_len"] = int(rng.choice([150, 400, 5000]))
        for k in ENV:
            os.environ.pop(k, None)
        if rng.random() < 0.35:
            os.environ["BSC_BAMDEV_REPLAY"] = "1"
        if rng.random() < 0.5:
            os.environ["BSC_BAMDEV_SLAB_KB"] = str(int(rng.choice([64, 128, 512])))
            os.environ["BSC_BAMDEV_PASS_KB"] = str(int(rng.choice([1, 16, 200])))
        try:
            want = TB.c_blocks(path, **kw)
        except BscError:
            want = None
        try:
            out = []
            with DeviceBamReader(caller, path, threads=int(rng.integers(1, 5)), **kw) as r:
                for tid, y, tpl, seq, ms in r.blocks():
                    out.append((tid, y, TE.templates_as_dicts(tpl, seq, ms)))
                cts, bases = r.filter_counts()
                n_replay += r.run_stats()["replay_passes"]
            got = (out, cts, bases)
        except BscError:
            got = None
        if (got is None) != (want is None) or (got is not None and got != want):
            print("DIFFERENCE at file %d (seed %d): sane=%s n=%d kw=%s env=%s" % (n_files, a.seed, sane, n, kw, {k: os.environ.get(k) for k in ENV}))
            W.write_bam(os.path.join(ROOT, "gpurun_out", "fuzz_bamdev_failing.bam"), TB.REFS, recs)
            sys.exit(1)
        n_files += 1
        if got is None:
            n_refused += 1
            continue
        n_blocks += len(got[0])
        n_tpl += sum(len(b[2]) for b in got[0])
        if n_files % 4 == 0 and sane:
            assert TB.py_blocks(path, **kw) == want, "py_bam differs from bamio.c"
            n_py += 1
        if n_files % 50 == 0:
            print("%d files, %d refused by both, %d blocks, %d templates, %d replay passes, %d files also against py_bam" % (n_files, n_refused, n_blocks, n_tpl, n_replay, n_py), flush=True)
print("fuzz_bamdev seed %d: %d files (%d refused by both readers), %d blocks, %d templates, %d one-lane replay passes, %d files also against oracle/py_bam.py: no difference"
      % (a.seed, n_files, n_refused, n_blocks, n_tpl, n_replay, n_py))
