"""Frame-level parity: our bitstream vs the reference's, byte for byte.

The reference side is (a) the golden SHA1s in tests/golden/golden.json, produced by oracle/_ref
(the real reference, compiled from /root/reference by oracle/Makefile) with tools/make_golden.py,
and (b) oracle/_ref run live when the binaries are present (they travel to the GPU box).

* `-m "not gpu"`: the kernels run through the CPU wave-emulation test build (tests/emu) -- this
  checks kernel logic + host entropy coding without a GPU.
* `-m gpu`: the same cases through libwelship.so on the MI355X (the product path).
"""
import hashlib
import json
import os
import subprocess

import pytest

import openh264_amd as oh
from openh264_amd.utils.synth import make_sequence, synth_sequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
SMALL = [k for k, v in GOLDEN.items() if v["w"] * v["h"] * v["frames"] <= 640 * 368 * 3]
LARGE = [k for k in GOLDEN if k not in SMALL]


def run_case(name, lib_path, ref_tools, tmp_path):
    g = GOLDEN[name]
    yuv = make_sequence(g.get("content", "synth"), g["w"], g["h"], g["frames"])
    assert hashlib.sha1(yuv).hexdigest() == g["input_sha1"], "synthetic generator changed"
    params = dict(fMaxFrameRate=30.0, iTargetBitrate=5000000)
    params.update(g["params"])
    stats = {}
    bs, recon = oh.encode_sequence(yuv, g["w"], g["h"], lib_path=lib_path, stats=stats, **params)
    assert len(bs) == g["bytes"]
    assert hashlib.sha1(bs).hexdigest() == g["sha1"]
    if name.endswith("_overflow"):      # the case must really go through the re-encode loop
        assert stats["overflow_reencodes"] > 0
    if name.endswith("_scene"):         # ... and this one must contain the IDR the scene-change detector inserts
        assert bs.count(b"\x00\x00\x00\x01\x65") == 2
    if ref_tools:
        fi, fo, fd = str(tmp_path / "in.yuv"), str(tmp_path / "ref.264"), str(tmp_path / "dec.yuv")
        open(fi, "wb").write(yuv)
        subprocess.check_call([ref_tools["enc"], "-i", fi, "-w", str(g["w"]), "-h", str(g["h"]), "-o", fo] + g["ref_flags"],
                              stdout=subprocess.DEVNULL)
        assert open(fo, "rb").read() == bs, "differs from oracle/_ref run live"
        # our reconstruction (after in-loop deblocking) must be what a decoder reconstructs
        ours = str(tmp_path / "ours.264")
        open(ours, "wb").write(bs)
        subprocess.check_call([ref_tools["dec"], ours, fd], stdout=subprocess.DEVNULL)
        dec = open(fd, "rb").read()
        fsz = g["w"] * g["h"] * 3 // 2
        assert dec[-fsz:] == recon, "device reconstruction differs from the decoder's"


@pytest.mark.parametrize("name", SMALL)
def test_emu_small(name, emu_lib, ref_tools, tmp_path):
    run_case(name, emu_lib, ref_tools, tmp_path)


@pytest.mark.parametrize("name", LARGE)
def test_emu_large(name, emu_lib, ref_tools, tmp_path):
    run_case(name, emu_lib, ref_tools, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_hip(name, hip_lib, ref_tools, tmp_path):
    run_case(name, hip_lib, ref_tools, tmp_path)


@pytest.mark.gpu
def test_hip_backend_is_gfx950(hip_lib):
    enc = oh.Encoder(hip_lib)
    p = enc.GetDefaultParams()
    p.iPicWidth, p.iPicHeight, p.iDLayerQp = 64, 64, 24
    assert enc.InitializeExt(p) == 0, enc.last_error()
    assert "gfx950" in enc.backend_name()
    enc.Uninitialize()


def test_unsupported_params_are_rejected(emu_lib):
    enc = oh.Encoder(emu_lib)
    p = enc.GetDefaultParams()
    p.iPicWidth, p.iPicHeight = 64, 64
    p.iRCMode = 1
    assert enc.InitializeExt(p) == oh.cmUnsupportedData
    p.iRCMode = -1
    p.iPicWidth = 8
    assert enc.InitializeExt(p) == oh.cmInitParaError
    p.iPicWidth = 64
    p.iEntropyCodingModeFlag = 1
    assert enc.InitializeExt(p) == oh.cmUnsupportedData


def test_product_library_fails_loudly_without_gpu(hip_lib):
    """libwelship.so must load and export the ABI anywhere, and refuse to run without an MI355X."""
    from conftest import has_gpu
    if has_gpu():
        pytest.skip("GPU present")
    enc = oh.Encoder(hip_lib)
    p = enc.GetDefaultParams()
    p.iPicWidth, p.iPicHeight = 64, 64
    assert enc.InitializeExt(p) == oh.ERR_NO_DEVICE
    assert "no usable device" in enc.last_error()


def _live_4k(lib_path, ref_tools, tmp_path):
    """Largest supported picture (4096x2304, 36864 MBs, 4 slices, IDR + P) against oracle/_ref run live."""
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    w, h = 4096, 2304
    yuv = synth_sequence(w, h, 2)
    bs, recon = oh.encode_sequence(yuv, w, h, lib_path=lib_path, iDLayerQp=28, uiIntraPeriod=0, fMaxFrameRate=30.0,
                                   iTargetBitrate=5000000, uiSliceMode=1, uiSliceNum=4)
    fi, fo, fd = str(tmp_path / "in.yuv"), str(tmp_path / "ref.264"), str(tmp_path / "dec.yuv")
    open(fi, "wb").write(yuv)
    subprocess.check_call([ref_tools["enc"], "-i", fi, "-w", str(w), "-h", str(h), "-o", fo, "-rc", "-1", "-qp", "28", "-fps", "30",
                           "-iper", "0", "-slcmd", "1", "-slcnum", "4", "-quiet"], stdout=subprocess.DEVNULL)
    assert open(fo, "rb").read() == bs
    subprocess.check_call([ref_tools["dec"], fo, fd], stdout=subprocess.DEVNULL)
    assert open(fd, "rb").read()[-w * h * 3 // 2:] == recon


def test_emu_4k(emu_lib, ref_tools, tmp_path):
    _live_4k(emu_lib, ref_tools, tmp_path)


@pytest.mark.gpu
def test_hip_4k(hip_lib, ref_tools, tmp_path):
    _live_4k(hip_lib, ref_tools, tmp_path)


@pytest.mark.gpu
def test_hip_4k_one_band_two_macroblocks_per_wave(hip_lib, ref_tools, tmp_path, monkeypatch):
    """The same picture deblocked as ONE band, two macroblocks per wavefront (k_deblock_pairs): 36864 macroblocks are more than the kernel keeps
    items for in LDS (WH_DB_ITEMS_LDS_MAX_MB), so this is the launch that reads the item list from device memory."""
    monkeypatch.setenv("WELSHIP_DB_WHOLE", "1")
    _live_4k(hip_lib, ref_tools, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("waves", ["6", "8", "12", "14", "16"])
def test_hip_wave_variants(waves, hip_lib):
    """The mode-decision pool runs with 16 waves per workgroup by default (k_inter_pool<1024>: one workgroup per CU); it is built for
    6 / 12 / 14 / 16 waves (<384>, <768>, <896>, <1024>: the LDS a launch has decides, csrc/hip/hip_backend.hip WH_LAUNCH_POOL) and every
    instantiation that ships is selected here once (8 takes <768>).  The count is forced (read once per process) and must give the same bits."""
    import sys
    name = "p_640x368_qp24_4slices"
    g = GOLDEN[name]
    code = ("import hashlib, sys; sys.path.insert(0, %r); import openh264_amd as oh; from openh264_amd.utils.synth import make_sequence, synth_sequence;"
            "yuv = synth_sequence(%d, %d, %d); bs, _ = oh.encode_sequence(yuv, %d, %d, lib_path=%r, fMaxFrameRate=30.0, iTargetBitrate=5000000, **%r);"
            "print(hashlib.sha1(bs).hexdigest())") % (ROOT, g["w"], g["h"], g["frames"], g["w"], g["h"], hip_lib, g["params"])
    env = dict(os.environ, WELSHIP_P_WAVES=waves)
    out = subprocess.check_output([sys.executable, "-c", code], env=env).decode().split()[-1]
    assert out == g["sha1"]


def test_band_order_is_topological():
    import ctypes
    import numpy as np
    lib = ctypes.CDLL(oh.build.build_emu())
    fn = lib.WelsHipDebugBuildMbOrder
    fn.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p]
    for mb_w, first, last, band in [(40, 0, 920, 6), (40, 13, 700, 3), (7, 0, 49, 2), (120, 240, 2280, 6), (5, 3, 4, 1)]:
        out = np.zeros(last - first, np.uint16)
        assert fn(mb_w, first, last, band, out.ctypes.data) == 0
        assert sorted(out.tolist()) == list(range(first, last))
        pos = {xy: i for i, xy in enumerate(out.tolist())}
        for xy in range(first, last):
            x = xy % mb_w
            for dep in ([xy - 1] if x > 0 else []) + [xy - mb_w] + ([xy - mb_w + 1] if x < mb_w - 1 else []) + ([xy - mb_w - 1] if x > 0 else []):
                if dep >= first:
                    assert pos[dep] < pos[xy]


def test_deblocking_pair_items_are_a_topological_order_of_independent_pairs():
    """The whole-picture deblocking order as items of one or two macroblocks (common/mb_order.h wh_build_db_pair_items, what k_deblock_pairs takes
    tickets for): every macroblock once; an item's macroblocks only depend on macroblocks of earlier items (deblocking dependencies: left, top,
    top-right -- deblocking.cpp:357-440 filters a macroblock's left and upper edge into its neighbours); a pair is two interior macroblocks of
    one 2:1 diagonal, on a diagonal of at least the asked length."""
    import ctypes
    import numpy as np
    from openh264_amd import build as whbuild
    lib = ctypes.CDLL(whbuild.build_emu())
    fn = lib.WelsHipDebugBuildDbPairItems
    fn.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p]
    for mb_w, mb_h, min_len in [(120, 68, 32), (120, 68, 2), (20, 12, 2), (7, 7, 2), (3, 9, 2), (1, 5, 2), (256, 144, 32), (80, 45, 1000)]:
        out = np.zeros(mb_w * mb_h + 1, np.uint32)
        assert fn(mb_w, mb_h, min_len, out.ctypes.data) == 0
        n = int(out[0])
        item_of, pairs = {}, 0
        for i in range(n):
            it = int(out[1 + i])
            a, pair = ((it >> 12) & 0xfff) * mb_w + (it & 0xfff), it >> 31           # x | y << 12 | pair << 31
            mbs = [a, a + mb_w - 2] if pair else [a]
            pairs += pair
            for xy in mbs:
                assert 0 <= xy < mb_w * mb_h and xy not in item_of
                item_of[xy] = i
            if pair:
                (ax, ay), (bx, by) = (a % mb_w, a // mb_w), (mbs[1] % mb_w, mbs[1] // mb_w)
                assert (bx, by) == (ax - 2, ay + 1)
                d = ax + 2 * ay
                assert sum(1 for y in range(mb_h) if 0 <= d - 2 * y < mb_w) >= min_len
                for x, y in ((ax, ay), (bx, by)):
                    assert 1 <= x <= mb_w - 2 and 1 <= y <= mb_h - 2
        assert len(item_of) == mb_w * mb_h
        for xy, i in item_of.items():
            x = xy % mb_w
            for dep in ([xy - 1] if x > 0 else []) + [xy - mb_w] + ([xy - mb_w + 1] if x < mb_w - 1 else []):
                if dep >= 0:
                    assert item_of[dep] < i
        if (mb_w, mb_h, min_len) == (120, 68, 32):
            assert pairs > 2500            # (three quarters of a 1080p picture's 8160 macroblocks go two at a time)
        if min_len == 1000 or mb_w < 5:
            assert pairs == 0


def test_session_lifecycle(emu_lib):
    """Uninitialize + InitializeExt on the same object behaves like a fresh object (no state leaks between sessions);
    ForceIntraFrame(false) is a successful no-op as in the reference (welsEncoderExt.cpp:487-500)."""
    def run(enc, w, h, n, qp, force_at=-1, idr=True, **kw):
        p = enc.GetDefaultParams()
        p.iPicWidth, p.iPicHeight, p.iDLayerQp, p.fMaxFrameRate, p.iTargetBitrate = w, h, qp, 30.0, 5000000
        for k, v in kw.items():
            setattr(p, k, v)
        assert enc.InitializeExt(p) == 0, enc.last_error()
        yuv, fsz, out = synth_sequence(w, h, n), w * h * 3 // 2, b""
        for f in range(n):
            if f == force_at:
                assert enc.ForceIntraFrame(idr) == 0
            rc, _, bs, _ = enc.EncodeFrame(yuv[f * fsz:(f + 1) * fsz], timestamp=f * 33)
            assert rc == 0
            out += bs
        return out
    e = oh.Encoder(emu_lib)
    a1 = run(e, 176, 144, 4, 24)
    e.Uninitialize()
    a2 = run(e, 320, 192, 3, 30, uiSliceMode=1, uiSliceNum=3)
    a3 = run(e, 176, 144, 4, 24)                   # re-initialised without Uninitialize
    a4 = run(e, 176, 144, 4, 24, force_at=2, idr=False)
    a5 = run(e, 176, 144, 4, 24, force_at=2, idr=True)
    e.Uninitialize()
    e.close()
    f = oh.Encoder(emu_lib)
    b2 = run(f, 320, 192, 3, 30, uiSliceMode=1, uiSliceNum=3)
    f.close()
    assert a1 == a3 == a4 and a2 == b2
    assert a5 != a1 and a5.count(b"\x00\x00\x00\x01\x65") == 2


def test_emu_deblock_per_edge(ref_tools, tmp_path):
    """kernels/deblock_mb.h filters a line's edges in registers, one pass per direction; the plain restatement (every edge a
    lane block of its own, byte accesses to the LDS tile) is kept behind WH_DB_PER_EDGE for A/B builds.  Both have to be
    bit-exact: the small golden cases with every deblocking variant."""
    from openh264_amd import build as B
    lib = B.build_emu(defines=("WH_DB_PER_EDGE",), tag="wh_db_per_edge")
    for name in SMALL:
        run_case(name, lib, ref_tools, tmp_path)


def test_emu_reverse_lane_order(ref_tools, tmp_path):
    """The wave emulation normally walks the lanes of a lane block from 0 to 63; a second test build walks them from 63 down.
    A lane block in which two lanes store to the same LDS word gives different results in the two -- and an undefined one on
    the GPU (found on the MI355X in round 2: the co-located state's padding word).  The golden cases must not care."""
    from openh264_amd import build as B
    lib = B.build_emu(defines=("WH_EMU_REVERSE",), tag="wh_emu_reverse")
    for name in SMALL:
        run_case(name, lib, ref_tools, tmp_path)


def test_emu_tiny_search_windows(ref_tools, tmp_path):
    """The search windows of a wave follow the search (kernels/inter_mb.h: wh_win_need, the diamond's step budget) and what they hold must
    never change a result.  A test build with windows that hold little more than the block (30 rows of luma, 12 of chroma) and a search start
    that asks for one sample of room reloads them all the time -- in mid-walk, before the refinement, for every skip prediction -- and has to
    write the golden streams too (incl. the 64-sample pan at level 1 and the 720p P picture)."""
    from openh264_amd import build as B
    lib = B.build_emu(defines=("WH_WIN_ROWS=30", "WH_WIN_MARGIN_Y=7", "WH_WIN_START=1", "WH_CWIN_ROWS=12"), tag="tiny_windows")
    for name in SMALL + ["p_1280x720_qp24"]:
        if name in GOLDEN:
            run_case(name, lib, ref_tools, tmp_path)


@pytest.mark.parametrize("case", [
    ("idr_interval_3_from_frame_4", ["-iper", "0", "-setidr", "4", "3"], [(4, oh.OPTION_IDR_INTERVAL, 3)], {}),
    ("idr_interval_off_from_frame_2", ["-iper", "2", "-setidr", "2", "-1"], [(2, oh.OPTION_IDR_INTERVAL, -1)], dict(uiIntraPeriod=2)),
    ("leave_all_idr_mode_at_frame_3", ["-iper", "1", "-setidr", "3", "4"], [(3, oh.OPTION_IDR_INTERVAL, 4)], dict(uiIntraPeriod=1)),
    ("complexity_high_from_frame_5", ["-iper", "0", "-setcplx", "5", "2"], [(5, oh.OPTION_COMPLEXITY, 2)], {}),
    ("complexity_low_from_frame_3", ["-iper", "0", "-complexity", "1", "-setcplx", "3", "0"], [(3, oh.OPTION_COMPLEXITY, 0)], dict(iComplexityMode=1)),
], ids=lambda c: c[0])
def test_set_option_matches_reference(case, emu_lib, ref_tools, tmp_path):
    """ISVCEncoder::SetOption in mid-stream (IDR interval, complexity) against the reference run live with the same calls."""
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    _, flags, options, extra = case
    w, h, n = 176, 144, 12
    yuv = synth_sequence(w, h, n)
    fi, fo = str(tmp_path / "in.yuv"), str(tmp_path / "ref.264")
    open(fi, "wb").write(yuv)
    subprocess.check_call([ref_tools["enc"], "-i", fi, "-w", str(w), "-h", str(h), "-o", fo, "-rc", "-1", "-qp", "28", "-quiet"] + flags,
                          stdout=subprocess.DEVNULL)
    params = dict(iDLayerQp=28, uiIntraPeriod=0, fMaxFrameRate=30.0, iTargetBitrate=5000000)
    params.update(extra)
    bs, _ = oh.encode_sequence(yuv, w, h, lib_path=emu_lib, options_at=options, **params)
    assert bs == open(fo, "rb").read()


def test_get_option(emu_lib):
    enc = oh.Encoder(emu_lib)
    p = enc.GetDefaultParams()
    p.iPicWidth, p.iPicHeight, p.uiIntraPeriod, p.iComplexityMode, p.fMaxFrameRate = 64, 64, 12, 1, 25.0
    assert enc.InitializeExt(p) == 0
    assert enc.GetOption(oh.OPTION_IDR_INTERVAL) == (0, 12) and enc.GetOption(oh.OPTION_COMPLEXITY) == (0, 1)
    assert enc.GetOption(oh.OPTION_DATAFORMAT) == (0, 23) and enc.GetOption(oh.OPTION_FRAME_RATE) == (0, 25.0)
    assert enc.SetOption(oh.OPTION_FRAME_RATE, 100.0) == 0 and enc.GetOption(oh.OPTION_FRAME_RATE) == (0, 60.0)
    assert enc.SetOption(oh.OPTION_IDR_INTERVAL, -5) == 0 and enc.GetOption(oh.OPTION_IDR_INTERVAL) == (0, 0)
    assert enc.SetOption(5, 1000) == oh.cmUnsupportedData          # ENCODER_OPTION_BITRATE: rate control is not part of this engine
    enc.close()


@pytest.mark.parametrize("at,iper,spsid", [(0, 0, 1), (3, 0, 1), (3, 2, 1), (5, 0, 0)])
def test_encode_parameter_sets_matches_reference(at, iper, spsid, emu_lib, ref_tools, tmp_path):
    """ISVCEncoder::EncodeParameterSets in mid-stream: SPS + PPS through the id strategy, later slices refer to the new ids."""
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    w, h, n = 176, 144, 8
    yuv = synth_sequence(w, h, n)
    fi, fo = str(tmp_path / "in.yuv"), str(tmp_path / "ref.264")
    open(fi, "wb").write(yuv)
    subprocess.check_call([ref_tools["enc"], "-i", fi, "-w", str(w), "-h", str(h), "-o", fo, "-rc", "-1", "-qp", "28", "-quiet",
                           "-iper", str(iper), "-spsid", str(spsid), "-paramsets", str(at)], stdout=subprocess.DEVNULL)
    bs, _ = oh.encode_sequence(yuv, w, h, lib_path=emu_lib, param_sets_at=at, iDLayerQp=28, uiIntraPeriod=iper, fMaxFrameRate=30.0,
                               iTargetBitrate=5000000, eSpsPpsIdStrategy=spsid)
    assert bs == open(fo, "rb").read()


@pytest.mark.parametrize("threads,flags,extra", [
    (4, ["-slcmd", "1", "-slcnum", "4"], dict(uiSliceMode=1, uiSliceNum=4)),
    (2, ["-slcmd", "1", "-slcnum", "3", "-deblock", "1"], dict(uiSliceMode=1, uiSliceNum=3, iLoopFilterDisableIdc=1)),
    (3, ["-slcmd", "2", "-slcmbnum", "50"], dict(uiSliceMode=2, uiSliceMbNum=[50] * 35)),
    (4, [], {}),
])
def test_matches_the_multithreaded_reference(threads, flags, extra, emu_lib, ref_tools, tmp_path):
    """iMultipleThreadIdc: this engine has no slice threads, but reproduces what the reference's slice threads do to the
    stream (deblocking across slice edges off) -- compared with the reference really running its slice threads."""
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    w, h, n = 320, 192, 4
    yuv = synth_sequence(w, h, n)
    fi, fo = str(tmp_path / "in.yuv"), str(tmp_path / "ref.264")
    open(fi, "wb").write(yuv)
    subprocess.check_call([ref_tools["enc"], "-i", fi, "-w", str(w), "-h", str(h), "-o", fo, "-rc", "-1", "-qp", "28", "-quiet", "-iper", "0",
                           "-threads", str(threads), "-loadbalancing", "0"] + flags, stdout=subprocess.DEVNULL)
    bs, _ = oh.encode_sequence(yuv, w, h, lib_path=emu_lib, iDLayerQp=28, uiIntraPeriod=0, fMaxFrameRate=30.0, iTargetBitrate=5000000,
                               iMultipleThreadIdc=threads, **extra)
    assert bs == open(fo, "rb").read()


def test_frame_info_metadata_matches_reference(emu_lib, ref_tools, tmp_path):
    """SFrameBSInfo as an application sees it (frame and layer types, NAL counts and lengths, ids, sub-sequence id, size,
    time stamp) against the reference's, frame by frame (ref_enc -dumpinfo)."""
    import ctypes as C
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    w, h, n = 176, 144, 7
    yuv = synth_sequence(w, h, n)
    fi, fm = str(tmp_path / "in.yuv"), str(tmp_path / "info.txt")
    open(fi, "wb").write(yuv)
    subprocess.check_call([ref_tools["enc"], "-i", fi, "-w", str(w), "-h", str(h), "-o", str(tmp_path / "r.264"), "-rc", "-1", "-qp", "28",
                           "-iper", "3", "-slcmd", "1", "-slcnum", "2", "-dumpinfo", fm, "-quiet"], stdout=subprocess.DEVNULL)
    enc = oh.Encoder(emu_lib)
    p = enc.GetDefaultParams()
    p.iPicWidth, p.iPicHeight, p.iDLayerQp, p.uiIntraPeriod, p.fMaxFrameRate, p.iTargetBitrate, p.uiSliceMode, p.uiSliceNum = w, h, 28, 3, 30.0, 5000000, 1, 2
    assert enc.InitializeExt(p) == 0
    fsz, out = w * h * 3 // 2, ""
    for f in range(n):
        buf = (C.c_uint8 * fsz).from_buffer_copy(yuv[f * fsz:(f + 1) * fsz])
        base = C.addressof(buf)
        pic = oh.SSourcePicture()
        pic.iColorFormat = 23
        pic.iStride[0], pic.iStride[1], pic.iStride[2] = w, w // 2, w // 2
        pic.pData[0], pic.pData[1], pic.pData[2] = base, base + w * h, base + w * h + (w // 2) * (h // 2)
        pic.iPicWidth, pic.iPicHeight, pic.uiTimeStamp = w, h, int(f * (1000.0 / 30) + 0.5)
        info = oh.SFrameBSInfo()
        assert enc._lib.WelsHipEncodeFrame(enc._h, C.byref(pic), C.byref(info)) == 0
        out += "frame %d type %d layers %d size %d ts %d\n" % (f, info.eFrameType, info.iLayerNum, info.iFrameSizeInBytes, info.uiTimeStamp)
        for li in range(info.iLayerNum):
            L = info.sLayerInfo[li]
            out += "  layer %d ltype %d ftype %d tid %d sid %d qid %d subseq %d nals" % (li, L.uiLayerType, L.eFrameType, L.uiTemporalId, L.uiSpatialId, L.uiQualityId, L.iSubSeqId)
            out += "".join(" %d" % L.pNalLengthInByte[k] for k in range(L.iNalCount)) + "\n"
    enc.close()
    assert out == open(fm).read()


def _whole_picture_band(lib, ref_tools, tmp_path, monkeypatch):
    """Large batches deblock a picture as ONE band (no seams between workgroups; hip_backend.hip run_deblock, WH_DB_WHOLE_TABLE)
    when the filter crosses slice edges anyway; WELSHIP_DB_WHOLE=1 forces that choice for a single picture: the multi-slice cases again."""
    monkeypatch.setenv("WELSHIP_DB_WHOLE", "1")
    # ... and there the pass filters TWO macroblocks per wavefront wherever a 2:1 diagonal has enough of them (common/mb_order.h
    # wh_build_db_pair_items; on the device k_deblock_pairs): these pictures' diagonals are short, so every diagonal pairs up here
    monkeypatch.setenv("WELSHIP_DB_PAIR_MIN", "2")
    names = [k for k in SMALL if GOLDEN[k]["params"].get("uiSliceNum", 1) > 1 or GOLDEN[k]["params"].get("uiSliceMode", 0) != 0]
    assert names
    for name in names[:6]:
        run_case(name, lib, ref_tools, tmp_path)


def test_emu_whole_picture_deblocking_band(emu_lib, ref_tools, tmp_path, monkeypatch):
    _whole_picture_band(emu_lib, ref_tools, tmp_path, monkeypatch)


@pytest.mark.gpu
def test_hip_whole_picture_deblocking_band(hip_lib, ref_tools, tmp_path, monkeypatch):
    _whole_picture_band(hip_lib, ref_tools, tmp_path, monkeypatch)


