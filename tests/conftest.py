import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def emu_lib():
    """CPU wave-emulation build of the kernel sources (test infrastructure, see tests/emu/)."""
    from openh264_amd import build as B
    return B.build_emu()


@pytest.fixture(scope="session")
def hip_lib():
    from openh264_amd import build as B
    return B.build_hip()


@pytest.fixture(scope="session")
def ref_tools():
    """Paths of the reference tools built by oracle/Makefile (None when not built)."""
    d = os.path.join(ROOT, "oracle", "_ref")
    enc, dec = os.path.join(d, "ref_enc"), os.path.join(d, "ref_dec")
    if os.path.exists(enc) and os.path.exists(dec):
        return {"enc": enc, "dec": dec, "dir": d}
    return None
