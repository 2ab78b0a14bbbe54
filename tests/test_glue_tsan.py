"""The overlapped glue's thread protocol (integration/amd_overlap_protocol.h — the code integration/call_genotypes_amd_overlap.c
is made of) under ThreadSanitizer, CPU only: integration/overlap_tsan.c runs it against the mock work_t with a print thread,
a meth profiling thread that reads work->ref1 and a process thread that overwrites ref1 the moment each call returns, over
stub bsc_* entries.  The harness must also CATCH round 2's bug (no wait for the profiling thread in a call that found no
block pending; reference: src/call_genotypes.c:244-251 waits in every call)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, name, *flags, src="overlap_tsan.c"):
    exe = str(tmp_path / name)
    cmd = ["gcc", "-O1", "-g", "-fsanitize=thread", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "integration"), *flags,
           os.path.join(ROOT, "integration", src), "-o", exe, "-lpthread"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0 and "tsan" in (p.stderr or "").lower():
        pytest.skip("this gcc has no ThreadSanitizer runtime: " + p.stderr[-200:])
    assert p.returncode == 0, p.stderr
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not found")
def test_overlap_protocol_is_tsan_clean(tmp_path):
    exe = _build(tmp_path, "overlap_tsan")
    for threshold in ("9000", "0", "1000000"):  # a handful of blocks per batch; every block its own batch; everything held until join
        p = subprocess.run([exe, "60", threshold], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        assert "ThreadSanitizer" not in p.stderr, p.stderr[-2000:]
        assert "record hash ok, reference hash ok" in p.stdout and " 0 saw ref1 change" in p.stdout
    # more tiny blocks than one bsc_blocks_submit_to call takes (65 536), under a threshold that never triggers: the glue must
    # flush by itself (the stub refuses a longer list, as the library does)
    p = subprocess.run([exe, "70000", "1000000000", "6"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ThreadSanitizer" not in p.stderr, p.stderr[-2000:]
    assert "70000 blocks" in p.stdout and "record hash ok, reference hash ok" in p.stdout


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not found")
def test_harness_catches_the_round2_protocol(tmp_path):
    exe = _build(tmp_path, "overlap_tsan_bug", "-DAMD_TEST_ROUND2_BUG")
    p = subprocess.run([exe, "60"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 or "ThreadSanitizer: data race" in p.stderr
    assert "mock_prepare_block" in p.stderr or "saw ref1 change" in p.stdout


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not found")
def test_bytes_form_protocol_is_tsan_clean(tmp_path):
    """INTEGRATION.md 2b (integration/amd_bcf_protocol.h, the code of integration/call_genotypes_amd_bcf.c): blocks queued in place, their
    BCF streams fetched one call later and handed to a writer thread in order; every third block longer than the first buffer"""
    exe = _build(tmp_path, "bcf_tsan", src="bcf_tsan.c")
    for n in ("1", "2", "61"):
        p = subprocess.run([exe, n], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        assert "ThreadSanitizer" not in p.stderr, p.stderr[-2000:]
        assert "stream hash ok" in p.stdout and " 0 saw ref1 change" in p.stdout
