import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure) -- builds oracle/libvote_oracle.so on first use."""
    from oracle import vote_oracle
    vote_oracle.lib()
    return vote_oracle


@pytest.fixture(scope="session")
def pkg():
    """The product package, registered under the importable name ``clean_pvnet_amd``."""
    import lib
    return lib._register_clean_pvnet_amd()


@pytest.fixture(scope="session")
def synth(pkg):
    from clean_pvnet_amd import synth as s
    return s


@pytest.fixture(scope="session")
def gpu():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch.device("cuda:0")
