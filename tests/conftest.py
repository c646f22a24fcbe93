import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu_lib():
    """The C ABI compiled for the host on the SIMT emulator (tests/emu) -- CPU-only kernel checks."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from animate_anything_amd import _lib
    return _lib.bind(build_emu.build())


@pytest.fixture()
def emu(emu_lib):
    from animate_anything_amd import _lib
    with _lib.use_library(emu_lib, host_pointers=True):
        yield emu_lib
