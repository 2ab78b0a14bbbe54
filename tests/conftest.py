import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import loader

    loader.build()
    return loader


@pytest.fixture(scope="session")
def tables(oracle):
    return oracle.Tables()


@pytest.fixture(scope="session")
def libm_exact(oracle):
    """True when the host's libm log/exp equal the bsmath.h replica bit for bit (glibc >= 2.28 on an x86-64
    CPU with FMA, where glibc selects its *_fma variants).  Probed, not assumed (oracle.loader.libm_exact)."""
    return oracle.libm_exact()


def pytest_report_header(config):
    """Says which oracle flavour the byte comparisons of this run are made against: with libm_exact the LIBM flavour —
    the reference's arithmetic on the host's own libm, independent of the product's bsmath.h — is asserted byte for
    byte; without it the LIBM comparison degrades to integers-exact + 1e-11 (tests/test_gpu_parity.py)."""
    try:
        from oracle import loader

        loader.build()
        return "bs_call_amd oracle: libm_exact=%s host=%s" % (loader.libm_exact(), loader.host_description())
    except Exception as e:  # the header must never break collection
        return "bs_call_amd oracle: libm_exact=unknown (%s)" % e


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The same line at the end of the run: `-q` (how the driver runs the suites) drops the header."""
    terminalreporter.write_line(pytest_report_header(config))
