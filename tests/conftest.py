import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of the tiers: cheap, wide evidence first (leaf primitives, frame parity, the reference's own clips and SHA1 tables), the randomised
# sessions last -- the driver runs the GPU tier with -x, and one late failure of a fuzzer must not blank the primitives' evidence
# (round 3: 43 of 116 GPU tests never ran behind a failing fuzz case).
_FILE_ORDER = ["test_abi", "test_prims_gpu", "test_leaf_gpu", "test_oracle_prims", "test_mb_order", "test_frame_parity", "test_reference_content", "test_downsample_gpu",
               "test_vaa", "test_frame_api_retry", "test_hooks_sha1", "test_hooks_screen", "test_hooks_simulcast", "test_hooks_cabac_threads",
               "test_multi_rank", "test_dropin_cli", "test_zz_dropin_gpu", "test_tools", "test_fuzz_parity", "test_hooks_dynslice"]


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _FILE_ORDER.index(name) if name in _FILE_ORDER else len(_FILE_ORDER) - 2
    items.sort(key=key)          # (stable: the order inside a file stays)


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
