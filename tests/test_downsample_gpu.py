"""Spatial down-sampling on the MI355X (include/welship.h 3b) against the reference's own C downsamplers
(codec/processing/src/downsample/downsamplefuncs.cpp:47-245, reached through oracle/_ref/libref_prims.so): bit-exact for the
sample region of every mode and for the layer pairs of BASELINE config 4 (1920x1080 -> 1280x720 general ratio, 1280x720 ->
640x360 and 640x360 -> 320x180 dyadic; chroma planes at half those sizes)."""
import ctypes as C
import os

import numpy as np
import pytest

import openh264_amd as oh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "oracle", "_ref", "libref_prims.so")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(SHIM), reason="oracle/_ref not built")]

HALF, QUARTER, THIRD, FAST, ACCURATE = 0, 1, 2, 3, 4
CASES = [
    (HALF, 1280, 720, 640, 360), (HALF, 640, 360, 320, 180), (HALF, 960, 540, 480, 270), (HALF, 322, 182, 161, 91), (HALF, 176, 144, 88, 72),
    (QUARTER, 1280, 720, 320, 180), (QUARTER, 644, 364, 161, 91),
    (THIRD, 1920, 1080, 640, 360), (THIRD, 960, 540, 320, 180),
    (FAST, 1920, 1080, 1280, 720), (FAST, 1280, 720, 854, 480), (FAST, 640, 360, 426, 246), (FAST, 333, 201, 150, 77),
    (ACCURATE, 960, 540, 640, 360), (ACCURATE, 640, 360, 427, 241), (ACCURATE, 320, 180, 213, 123), (ACCURATE, 167, 101, 75, 39),
]


@pytest.mark.parametrize("mode,sw,sh,dw,dh", CASES)
def test_downsample_matches_the_reference(hip_lib, mode, sw, sh, dw, dh):
    lib = oh.load_library(hip_lib)
    ref = C.CDLL(SHIM)
    rng = np.random.default_rng(sw * 131 + dh * 7 + mode)
    ss, ds = (sw + 63) // 64 * 64 + 64, (dw + 31) // 32 * 32 + 32
    src = rng.integers(0, 256, size=(sh + 2, ss), dtype=np.uint8)
    if mode in (FAST, ACCURATE):       # smooth content too: exercises the rounding of the weights, not just the clamps
        yy, xx = np.mgrid[0:sh + 2, 0:ss]
        src = ((src.astype(np.int32) // 8) + (xx * 3 + yy * 5) % 224).astype(np.uint8)
    a = np.full((dh, ds), 0xA5, dtype=np.uint8)
    b = a.copy()
    p8 = C.POINTER(C.c_uint8)
    ref.ref_downsample(mode, a.ctypes.data_as(p8), ds, dw, dh, src.ctypes.data_as(p8), ss, sw, sh)
    lib.WelsHipPrimDownsample.argtypes = [C.c_int, p8, C.c_int32, C.c_int32, C.c_int32, p8, C.c_int32, C.c_int32, C.c_int32]
    rc = lib.WelsHipPrimDownsample(mode, b.ctypes.data_as(p8), ds, dw, dh, src.ctypes.data_as(p8), ss, sw, sh)
    assert rc == 0
    assert np.array_equal(a[:, :dw], b[:, :dw])
    assert np.all(b[:, dw:] == 0xA5)              # nothing outside the destination rectangle is written
