"""The BAM reader's helper threads (csrc/bamio.c: BGZF blocks inflated ahead of the parser by up to 16 workers) under
ThreadSanitizer, CPU only: integration/bam_tsan.c compiled together with bamio.c and prep.c reads a file of a few hundred BGZF
blocks with 0 / 1 / 3 / 8 helpers, closes readers in mid-file, and must deliver the same bytes every time without a report."""
import importlib.util
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("make_bam", os.path.join(ROOT, "tools", "make_bam.py"))
W = importlib.util.module_from_spec(spec)
spec.loader.exec_module(W)


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not found")
def test_threaded_bgzf_reader_is_tsan_clean(tmp_path):
    exe = str(tmp_path / "bam_tsan")
    src = [os.path.join(ROOT, "integration", "bam_tsan.c"), os.path.join(ROOT, "bs_call_amd", "csrc", "bamio.c"),
           os.path.join(ROOT, "bs_call_amd", "csrc", "prep.c")]
    p = subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-fsanitize=thread", "-I" + os.path.join(ROOT, "include"), *src, "-o", exe, "-lz", "-lpthread", "-lm"],
                       capture_output=True, text=True)
    if p.returncode != 0 and "tsan" in (p.stderr or "").lower():
        pytest.skip("this gcc has no ThreadSanitizer runtime: " + p.stderr[-200:])
    assert p.returncode == 0, p.stderr[-3000:]
    rng = np.random.default_rng(5)
    n = 200_000
    ref = rng.integers(1, 5, n).astype(np.uint8)
    recs = W.wgbs_records(rng, ref, 0, 6000)
    bam = str(tmp_path / "t.bam")
    W.write_bam(bam, [("chr1", n)], recs, block=6000)  # small BGZF blocks: a few hundred of them, records cut by block ends
    r = subprocess.run([exe, bam], capture_output=True, text=True, timeout=600)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert "early closes ok" in r.stdout
