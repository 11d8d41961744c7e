"""the pre-analysis statistics on the device (WelsHipFrameVaa, kernels/vaa_pic.h) against the reference's C functions
VAACalcSad_c / VAACalcSadBgd_c / VAACalcSadSsd_c / VAACalcSadVar_c / VAACalcSadSsdBgd_c (codec/processing/src/vaacalc/vaacalcfuncs.cpp),
called straight out of oracle/_ref/libref_openh264.so: every result array, every flag combination CVAACalculation::Process
distinguishes, picture sizes that are no multiple of 16, a sequence of pictures (the earlier picture is then found on the device)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFLIB = os.path.join(ROOT, "oracle", "_ref", "libref_openh264.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REFLIB), reason="oracle/_ref not built")


class FrameCfg(C.Structure):
    _fields_ = [("iDevice", C.c_int32), ("iPicWidth", C.c_int32), ("iPicHeight", C.c_int32), ("iNumPictures", C.c_int32)]


class VaaJob(C.Structure):
    _fields_ = [("pCur", C.c_void_p * 3), ("iCurStride", C.c_int32 * 3), ("pRef", C.c_void_p * 3), ("iRefStride", C.c_int32 * 3),
                ("iPicWidth", C.c_int32), ("iPicHeight", C.c_int32), ("bCalcVar", C.c_int32), ("bCalcBgd", C.c_int32), ("bCalcSsd", C.c_int32),
                ("pSad8x8", C.c_void_p), ("pSsd16x16", C.c_void_p), ("pSum16x16", C.c_void_p), ("pSumOfSquare16x16", C.c_void_p),
                ("pSumOfDiff8x8", C.c_void_p), ("pMad8x8", C.c_void_p), ("pFrameSad", C.c_void_p)]


def _picture(rng, w16, h16, stride):
    y = rng.integers(0, 256, size=(h16 + 1, stride), dtype=np.uint8)
    u = rng.integers(0, 256, size=(h16 // 2 + 1, stride // 2), dtype=np.uint8)
    v = rng.integers(0, 256, size=(h16 // 2 + 1, stride // 2), dtype=np.uint8)
    return [np.ascontiguousarray(a) for a in (y, u, v)]


def _reference(ref, cur, prev, w, h, stride, var, bgd, ssd):
    n = (w >> 4) * (h >> 4)
    P = C.c_void_p
    out = {"sad": np.full(4 * n, -7, np.int32), "sd": np.full(4 * n, -7, np.int32), "mad": np.full(4 * n, 201, np.uint8),
           "sum": np.full(n, -7, np.int32), "sq": np.full(n, -7, np.int32), "ssd": np.full(n, -7, np.int32)}
    fs = C.c_int32(0)
    a = [cur[0].ctypes.data, prev[0].ctypes.data, w, h, stride, C.addressof(fs)]
    p = {k: v.ctypes.data for k, v in out.items()}
    if bgd and ssd:
        f = getattr(ref, "_ZN6WelsVP18VAACalcSadSsdBgd_cEPKhS1_iiiPiS2_S2_S2_S2_S2_Ph"); f.argtypes = [P, P, C.c_int, C.c_int, C.c_int, P] + [P] * 6
        f(*a, p["sad"], p["sum"], p["sq"], p["ssd"], p["sd"], p["mad"])
    elif bgd:
        f = getattr(ref, "_ZN6WelsVP15VAACalcSadBgd_cEPKhS1_iiiPiS2_S2_Ph"); f.argtypes = [P, P, C.c_int, C.c_int, C.c_int, P] + [P] * 3
        f(*a, p["sad"], p["sd"], p["mad"])
    elif ssd:
        f = getattr(ref, "_ZN6WelsVP15VAACalcSadSsd_cEPKhS1_iiiPiS2_S2_S2_S2_"); f.argtypes = [P, P, C.c_int, C.c_int, C.c_int, P] + [P] * 4
        f(*a, p["sad"], p["sum"], p["sq"], p["ssd"])
    elif var:
        f = getattr(ref, "_ZN6WelsVP15VAACalcSadVar_cEPKhS1_iiiPiS2_S2_S2_"); f.argtypes = [P, P, C.c_int, C.c_int, C.c_int, P] + [P] * 3
        f(*a, p["sad"], p["sum"], p["sq"])
    else:
        f = getattr(ref, "_ZN6WelsVP12VAACalcSad_cEPKhS1_iiiPiS2_"); f.argtypes = [P, P, C.c_int, C.c_int, C.c_int, P, P]
        f(*a, p["sad"])
    return out, fs.value


def _device(lib, ctx, cur, prev, w, h, stride, var, bgd, ssd):
    n = (w >> 4) * (h >> 4)
    out = {"sad": np.full(4 * n, -7, np.int32), "sd": np.full(4 * n, -7, np.int32), "mad": np.full(4 * n, 201, np.uint8),
           "sum": np.full(n, -7, np.int32), "sq": np.full(n, -7, np.int32), "ssd": np.full(n, -7, np.int32)}
    fs = C.c_int32(0)
    j = VaaJob()
    for i in range(3):
        j.pCur[i] = cur[i].ctypes.data; j.iCurStride[i] = stride if i == 0 else stride // 2
        j.pRef[i] = prev[i].ctypes.data; j.iRefStride[i] = stride if i == 0 else stride // 2
    j.iPicWidth, j.iPicHeight, j.bCalcVar, j.bCalcBgd, j.bCalcSsd = w, h, var, bgd, ssd
    j.pSad8x8, j.pSsd16x16, j.pSum16x16, j.pSumOfSquare16x16 = out["sad"].ctypes.data, out["ssd"].ctypes.data, out["sum"].ctypes.data, out["sq"].ctypes.data
    j.pSumOfDiff8x8, j.pMad8x8, j.pFrameSad = out["sd"].ctypes.data, out["mad"].ctypes.data, C.addressof(fs)
    rc = lib.WelsHipFrameVaa(ctx, C.byref(j))
    assert rc == 0, rc            # (widths that are no multiple of 16 too: the C functions' skewed walk, kernels/vaa_pic.h wh_vaa_mb_skewed)
    return out, fs.value


def _check(libpath):
    lib = C.CDLL(libpath)
    ref = C.CDLL(REFLIB)
    lib.WelsHipFrameCtxCreate.argtypes = [C.POINTER(C.c_void_p), C.POINTER(FrameCfg)]
    lib.WelsHipFrameVaa.argtypes = [C.c_void_p, C.POINTER(VaaJob)]
    lib.WelsHipFrameCtxDestroy.argtypes = [C.c_void_p]
    rng = np.random.default_rng(5)
    for (w, h) in [(320, 192), (176, 144), (338, 250), (320, 180), (1280, 720), (200, 100), (185, 97), (1001, 563)]:
        w16, h16 = (w + 15) // 16 * 16, (h + 15) // 16 * 16
        stride = w16 + 64
        ctx = C.c_void_p()
        cfg = FrameCfg(0, w16, h16, 3)
        assert lib.WelsHipFrameCtxCreate(C.byref(ctx), C.byref(cfg)) == 0
        pics = [_picture(rng, w16, h16, stride) for _ in range(4)]
        # a chain of pictures (the earlier one resident from the call before), then every flag combination on fresh pairs
        order = [(1, 0), (2, 1), (3, 2), (0, 3), (2, 0)]
        flags = [(0, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 0), (0, 1, 1), (1, 1, 0)]
        for k, (ci, pi) in enumerate(order):
            for (var, bgd, ssd) in (flags if k >= 3 else [flags[k % 2]]):
                want, wfs = _reference(ref, pics[ci], pics[pi], w, h, stride, var, bgd, ssd)
                got, gfs = _device(lib, ctx, pics[ci], pics[pi], w, h, stride, var, bgd, ssd)
                assert gfs == wfs, (w, h, var, bgd, ssd, gfs, wfs)
                for name in want:
                    assert np.array_equal(want[name], got[name]), (w, h, var, bgd, ssd, name)
            pics[ci][0][3, 5] ^= 0x40        # the host buffer changes before it is analysed again: the device copy must follow
        lib.WelsHipFrameCtxDestroy(ctx)


def test_vaa_statistics_on_emulation(emu_lib):
    _check(emu_lib)


@pytest.mark.gpu
def test_vaa_statistics_on_the_mi355x(hip_lib):
    _check(hip_lib)


def _hooked_sessions_of_odd_widths(lib, tmp_path):
    """Through the dispatch-table binding: sessions whose width is no multiple of 16 keep their pre-analysis on the device (the hook runs the
    reference's function as well and compares every array: WELS_HIP_CHECK_VAA=1), and the streams are the reference's."""
    import subprocess
    from openh264_amd.utils.synth import synth_sequence
    refdir = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.exists(os.path.join(refdir, "ref_enc_hip")):
        pytest.skip("oracle/_ref (hooked reference) not built")
    for (w, h) in ((200, 100), (185, 97)):
        src = str(tmp_path / "c.yuv")
        open(src, "wb").write(synth_sequence(w, h, 5))
        outs = []
        for exe, env in (("ref_enc", dict(os.environ)), ("ref_enc_hip", dict(os.environ, WELSHIP_LIB=lib, WELS_HIP_TRACE="1", WELS_HIP_CHECK_VAA="1")),
                         ("ref_enc_hip", dict(os.environ, WELSHIP_LIB=lib, WELS_HIP_TRACE="1"))):
            out = str(tmp_path / (exe + ".264"))
            p = subprocess.run([os.path.join(refdir, exe), "-i", src, "-w", str(w), "-h", str(h), "-o", out, "-quiet", "-rc", "1", "-bitrate", "200000", "-bgd", "1", "-scene", "1"],
                               env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            err = p.stderr.decode(errors="replace")
            assert p.returncode == 0, err[-1500:]
            if "WELS_HIP_CHECK_VAA" in env:
                assert err.count("the device's pre-analysis equals the reference's") >= 4 and "stays on the host" not in err, err[-1500:]
            elif exe == "ref_enc_hip":
                assert err.count("pre-analysis statistics of layer 0 on the device") >= 4, err[-1500:]
            outs.append(open(out, "rb").read())
        assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 500


def test_hooked_sessions_of_odd_widths_on_emulation(emu_lib, tmp_path):
    _hooked_sessions_of_odd_widths(emu_lib, tmp_path)


@pytest.mark.gpu
def test_hooked_sessions_of_odd_widths_on_the_mi355x(hip_lib, tmp_path):
    _hooked_sessions_of_odd_widths(hip_lib, tmp_path)


def _hooked_background_detection(lib, tmp_path):
    """CWelsPreProcess::BackgroundDetection on the device (WelsHipFrameBgd, kernels/bgd_pic.h: the in-place raster pass of
    BackgroundDetection.cpp:333-374 walked diagonal by diagonal), from the statistics the pre-analysis call left there.  WELS_HIP_CHECK_BGD=1
    makes the hook run the VP library's function as well and compare every macroblock's flag; without it the streams must be the reference's."""
    import subprocess
    from openh264_amd.utils.synth import synth_sequence
    refdir = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.exists(os.path.join(refdir, "ref_enc_hip")):
        pytest.skip("oracle/_ref (hooked reference) not built")
    clips = []
    src = str(tmp_path / "s.yuv")
    open(src, "wb").write(synth_sequence(320, 192, 8))
    clips.append((src, 320, 192))
    cam = os.path.join(refdir, "res", "CiscoVT2people_320x192_12fps.yuv")
    if os.path.exists(cam):
        clips.append((cam, 320, 192))
    for (src, w, h) in clips:
        outs = []
        for exe, env in (("ref_enc", dict(os.environ)), ("ref_enc_hip", dict(os.environ, WELSHIP_LIB=lib, WELS_HIP_TRACE="1", WELS_HIP_CHECK_BGD="1")),
                         ("ref_enc_hip", dict(os.environ, WELSHIP_LIB=lib, WELS_HIP_TRACE="1"))):
            out = str(tmp_path / (exe + ".264"))
            p = subprocess.run([os.path.join(refdir, exe), "-i", src, "-w", str(w), "-h", str(h), "-o", out, "-quiet", "-rc", "1", "-bitrate", "300000", "-bgd", "1",
                                "-scene", "1", "-frames", "24"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            err = p.stderr.decode(errors="replace")
            assert p.returncode == 0 and "differs" not in err, err[-1500:]
            if "WELS_HIP_CHECK_BGD" in env:
                assert err.count("the device's background detection equals the reference's") >= 5, err[-1500:]
            elif exe == "ref_enc_hip":
                assert err.count("background detection of layer 0 on the device") >= 5 and "background detection of layer 0 stays" not in err, err[-1500:]
            outs.append(open(out, "rb").read())
        assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 500


def test_hooked_background_detection_on_emulation(emu_lib, tmp_path):
    _hooked_background_detection(emu_lib, tmp_path)


@pytest.mark.gpu
def test_hooked_background_detection_on_the_mi355x(hip_lib, tmp_path):
    _hooked_background_detection(hip_lib, tmp_path)
