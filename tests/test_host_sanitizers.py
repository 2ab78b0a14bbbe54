"""The library's HOST C code (csrc/bamio.c, bamstream.c, inflate_fast.c, prep.c, dbsnp.c, bcf.c, report.c, vcf_format.c, refseq.c, bscall_api.c's argument
checks) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (`make san`: gcc's runtimes; GPU sanitizers are not
available on the pool): (1) the host-layer tests of this directory once more against the sanitized build, (2) the
corrupted-input fuzz of the readers (tools/fuzz_host_inputs.py: damaged BAM / SAM / FASTA / dbSNP files must be read or
refused with an error).  Neither may produce a sanitizer report.  What the first runs of this found is in DESIGN.md section 3."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_LIB = os.path.join(ROOT, "bs_call_amd", "lib", "san", "libbscall_amd_san.so")


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True)
    path = p.stdout.strip()
    return path if p.returncode == 0 and os.path.isabs(path) and os.path.exists(path) else None


@pytest.fixture(scope="module")
def san_env():
    if shutil.which("gcc") is None or shutil.which("make") is None:
        pytest.skip("gcc / make not found")
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("this gcc has no ASan / UBSan runtime")
    if not os.path.exists(os.path.join(ROOT, "bs_call_amd", "lib", "libbscall_amd.so")):
        pytest.skip("library not built (python -c 'import __graft_entry__ as g; g.build()')")
    p = subprocess.run(["make", "-s", "san"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and os.path.exists(SAN_LIB), p.stdout[-2000:] + p.stderr[-2000:]
    return dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1",
                BSCALL_AMD_LIB=SAN_LIB)


def _clean(p):
    out = p.stdout + p.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert p.returncode == 0, out[-4000:]


def test_host_layer_tests_under_asan_ubsan(san_env):
    files = ["test_bam.py", "test_prep.py", "test_dbsnp.py", "test_bcf.py", "test_report.py", "test_abi_exports.py", "test_bamstream.py", "test_inflate_fast.py"]
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *[os.path.join(ROOT, "tests", f) for f in files]],
                       cwd=ROOT, env=san_env, capture_output=True, text=True, timeout=1500)
    _clean(p)
    assert " passed" in p.stdout


def test_loaded_library_is_the_sanitized_one(san_env):
    code = ("import bs_call_amd._lib as L; L.load(); m = open('/proc/self/maps').read(); "
            "assert 'libbscall_amd_san.so' in m and 'libasan' in m, m[-500:]; print('ok')")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=san_env, capture_output=True, text=True, timeout=300)
    _clean(p)


@pytest.mark.parametrize("seed", [1, 2])
def test_damaged_inputs_are_read_or_refused(san_env, seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_host_inputs.py"), "--seed", str(seed), "--rounds", "600"], cwd=ROOT,
                       env=san_env, capture_output=True, text=True, timeout=1500)
    _clean(p)
    assert "every reader finished or refused" in p.stdout
