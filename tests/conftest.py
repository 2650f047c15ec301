import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle

    oracle.build()
    oracle.load()
    return oracle


@pytest.fixture(scope="session")
def abi_mod():
    from tbraymarcherplugin_amd import abi

    abi.load()
    return abi


@pytest.fixture(scope="session")
def gpu(abi_mod):
    """The HIP path must be the one that runs: no device or no extension is a failure, not a skip."""
    n = abi_mod.device_count()
    assert n > 0, "gpu-marked test on a machine without a HIP device: " + abi_mod.load().tbrm_last_error().decode()
    return abi_mod


@pytest.fixture
def tunables(abi_mod):
    """tunables(name, value) flips one of the library's process-wide A/B switches (tbrm_set_tunable) for the duration of
    the test; every switch touched goes back to what it was."""
    saved = {}

    def set_(name, value):
        if name not in saved:
            saved[name] = abi_mod.get_tunable(name)
        abi_mod.set_tunable(name, value)

    yield set_
    for name, value in saved.items():
        abi_mod.set_tunable(name, value)


def small_volume(dims, dtype, seed=0x5EED0002):
    from tbraymarcherplugin_amd import synthetic

    return synthetic.make_volume_numpy(dims, dtype, seed)
