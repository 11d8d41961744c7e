"""The C-ABI library loads without a GPU and exports every symbol include/welship.h declares."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header(hip_lib):
    hdr = open(os.path.join(ROOT, "include", "welship.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(WelsHip[A-Za-z0-9]+)\s*\(", hdr))
    assert len(names) >= 30
    lib = C.CDLL(hip_lib)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_no_device_is_loud(hip_lib):
    from conftest import has_gpu
    if has_gpu():
        return
    lib = C.CDLL(hip_lib)
    out = (C.c_int32 * 1)()
    buf = (C.c_uint8 * 1024)()
    off = (C.c_int32 * 1)(0)
    rc = lib.WelsHipPrimSampleSad(0, 1, buf, C.c_size_t(1024), 32, off, buf, C.c_size_t(1024), 32, off, out)
    assert rc == 100      # WELSHIP_ERR_NO_DEVICE
