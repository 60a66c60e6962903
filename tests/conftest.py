import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMU_PATH = os.path.join(ROOT, "tests", "hostemu", "libcamcal_hostemu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def use_emulator(lib):
    """Route every call of the package to the given library (the CPU lane emulator) on the CPU by replacing
    api._resolve from the outside -- the product itself has no such route.  Returns the undo function."""
    import torch
    from cameracalibration_amd import api
    orig = api._resolve
    api._resolve = lambda: (lib, torch.device("cpu"))
    return lambda: setattr(api, "_resolve", orig)


@pytest.fixture(scope="session")
def _emu_lib():
    if not os.path.exists(EMU_PATH):
        subprocess.run(["make", "-C", ROOT, "hostemu"], check=True)
    from cameracalibration_amd import _lib
    return _lib.load(EMU_PATH)


@pytest.fixture()
def emu(_emu_lib):
    """CPU lane emulator of libcamcal.so (test infrastructure, see tests/hostemu).  While a test
    holds this fixture every call of the package runs on it (use_emulator above); explicit
    Engine(..., lib=emu) constructions work as before."""
    undo = use_emulator(_emu_lib)
    yield _emu_lib
    undo()


@pytest.fixture(scope="session")
def oracle_c():
    path = os.path.join(ROOT, "oracle", "liboracle_c.so")
    if not os.path.exists(path):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    from oracle import cport
    return cport
