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
    CPU with FMA, where glibc selects its *_fma variants).  Probed, not assumed."""
    import numpy as np

    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(1e-5, 10, 200_000), rng.uniform(0.93, 1.07, 50_000)])
    y = rng.uniform(-745, 30, 250_000)
    ok = (oracle.log_array(x, 0).view(np.int64) == oracle.log_array(x, 1).view(np.int64)).all() and (
        oracle.exp_array(y, 0).view(np.int64) == oracle.exp_array(y, 1).view(np.int64)
    ).all()
    return bool(ok)
