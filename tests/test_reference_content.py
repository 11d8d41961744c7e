"""Parity on the reference's own test clips and pinning of the oracle against the reference's checked-in hashes.

The clips (res/*.264 decoded with the reference decoder, res/*.yuv) and the stock configuration files are copied to
oracle/_ref/res/ by oracle/Makefile, so the GPU box -- where /root/reference does not exist -- runs the same cases on
libwelship.so (`-m gpu`); the CPU tier runs them on the wave emulation of the kernel sources.

Hashes:
  bb6dba56...  1080p, 54 frames, 4 slices, QP 24, P frames: stock h264enc of the reference
  02de34fb...  720p, 300 frames, all-IDR, QP 24 (BASELINE config 2's stand-in clip)
  dd643761...  BASELINE config 1: welsenc.cfg on CiscoVT2people_160x96 with -rc -1 -lqp 0 24
  08ade185..., 672a52fb..., e60f12e3...   test/api/encoder_test.cpp:104-115 (EncoderOutputTest, entries 1-3)
  81bde26c...  test/encoder_binary_comparison/SHA1Table/BA_MW_D.264_AllCases_SHA1_Table.csv row 1
"""
import hashlib
import os
import subprocess

import pytest

import openh264_amd as oh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = os.path.join(ROOT, "oracle", "_ref", "res")
pytestmark = pytest.mark.skipif(not os.path.isdir(RES), reason="oracle/_ref/res not built (oracle/Makefile)")


def _decode(ref_tools, name, tmp_path):
    out = str(tmp_path / (name + ".yuv"))
    subprocess.check_call([ref_tools["dec"], os.path.join(RES, name), out], stdout=subprocess.DEVNULL)
    return open(out, "rb").read()


def _ref_encode(ref_tools, yuv, w, h, flags, tmp_path):
    fi, fo = str(tmp_path / "in.yuv"), str(tmp_path / "ref.264")
    open(fi, "wb").write(yuv)
    subprocess.check_call([ref_tools["enc"], "-i", fi, "-w", str(w), "-h", str(h), "-o", fo, "-rc", "-1", "-fps", "30", "-quiet"] + flags,
                          stdout=subprocess.DEVNULL)
    return open(fo, "rb").read()


def _sha1(b):
    return hashlib.sha1(b).hexdigest()


# ---- the oracle itself, pinned to the reference's checked-in golden hashes (no engine involved) -------------------------
API_GOLDEN = [  # test/api/encoder_test.cpp:104-115: file, usage, w, h, fps, hash (SEncParamBase: 5 Mbps, RC quality mode)
    ("CiscoVT2people_160x96_6fps.yuv", 160, 96, 6.0, "08ade1853e4e49d50be675393780e75519586143"),
    ("CiscoVT2people_320x192_12fps.yuv", 320, 192, 12.0, "672a52fb6b6e6d52b5b3f3480d13d44e88481fb9"),
    ("Static_152_100.yuv", 152, 100, 6.0, "e60f12e3c24500d4306d812b0811d3c21855dd1c"),
]


@pytest.mark.parametrize("name,w,h,fps,sha", API_GOLDEN)
def test_oracle_reproduces_api_golden_hash(ref_tools, tmp_path, name, w, h, fps, sha):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    fo = str(tmp_path / "o.264")
    subprocess.check_call([ref_tools["enc"], "-i", os.path.join(RES, name), "-w", str(w), "-h", str(h), "-o", fo, "-base", "-rc", "0",
                           "-fps", str(fps), "-quiet"], stdout=subprocess.DEVNULL)
    assert _sha1(open(fo, "rb").read()) == sha


def test_oracle_reproduces_sha1_table_row_1(ref_tools, tmp_path):
    """Row 1 of BA_MW_D.264_AllCases_SHA1_Table.csv, run the way run_BinarySHA1Comparison.sh:165-200 runs it: the stock
    h264enc front-end with welsenc.cfg + layer2.cfg copied as layer0..3.cfg and the row's options on the command line."""
    h264enc = os.path.join(ROOT, "oracle", "_ref", "h264enc_ref")
    if not ref_tools or not os.path.exists(h264enc):
        pytest.skip("oracle/_ref not built")
    yuv = _decode(ref_tools, "BA_MW_D.264", tmp_path)
    assert _sha1(yuv) == "afd7a9765961ca241bb4bdf344b31397bec7465a"          # the table's InputYUVSHA1 column
    for k in range(4):
        open(str(tmp_path / ("layer%d.cfg" % k)), "wb").write(open(os.path.join(RES, "layer2.cfg"), "rb").read())
    open(str(tmp_path / "welsenc.cfg"), "wb").write(open(os.path.join(RES, "welsenc.cfg"), "rb").read())
    src = str(tmp_path / "BA_MW_D.264.yuv")
    row = ("-utype 0 -frms 50 -numl 1 -numtl 1 -sw 176 -sh 144 -dw 0 176 -dh 0 144 -dw 1 0 -dh 1 0 -dw 2 0 -dh 2 0 -dw 3 0 -dh 3 0 "
           "-frout 0 30 -frout 1 30 -frout 2 30 -frout 3 30 -lqp 0 26 -lqp 1 26 -lqp 2 26 -lqp 3 26 -rc 1 -fs 1 -tarb 300.00 "
           "-ltarb 0 300.00 -ltarb 1 0 -ltarb 2 0 -ltarb 3 0 -lmaxb 0 300.00 -lmaxb 1 0 -lmaxb 2 0 -lmaxb 3 0 "
           "-slcmd 0 0 -slcnum 0 0 -slcmd 1 0 -slcnum 1 0 -slcmd 2 0 -slcnum 2 0 -slcmd 3 0 -slcnum 3 0 -nalsize 0 -iper 0 "
           "-thread 1 -loadbalancing 0 -ltr 0 -db 0 -denois 1 -scene 0 -bgd 0 -aq 0").split()
    out = str(tmp_path / "t.264")
    subprocess.check_call([h264enc, "welsenc.cfg", "-lconfig", "0", "layer0.cfg", "-lconfig", "1", "layer1.cfg", "-lconfig", "2", "layer2.cfg",
                           "-lconfig", "3", "layer3.cfg", "-bf", out, "-org", src] + row, cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert _sha1(open(out, "rb").read()) == "81bde26c3836e50cdf57d0ff5c13f83e5964ca0a"


# ---- engine vs reference on the reference's clips -----------------------------------------------------------------------
def _run_1080p(lib, ref_tools, tmp_path):
    yuv = _decode(ref_tools, "VID_1920x1080_cavlc_temporal_direct.264", tmp_path)
    assert len(yuv) == 167961600                                   # 54 frames
    ref = _ref_encode(ref_tools, yuv, 1920, 1080, ["-iper", "0", "-qp", "24", "-slcmd", "1", "-slcnum", "4"], tmp_path)
    assert _sha1(ref) == "bb6dba56327985bc94c60ca1e7a004b4fb543513"
    bs, _ = oh.encode_sequence(yuv, 1920, 1080, lib_path=lib, iDLayerQp=24, uiIntraPeriod=0, fMaxFrameRate=30.0,
                               iTargetBitrate=5000000, uiSliceMode=1, uiSliceNum=4)
    assert bs == ref


def _run_720p_intra(lib, ref_tools, tmp_path, frames):
    yuv = _decode(ref_tools, "VID_1280x720_cavlc_temporal_direct.264", tmp_path)[: 1280 * 720 * 3 // 2 * frames]
    ref = _ref_encode(ref_tools, yuv, 1280, 720, ["-iper", "1", "-qp", "24"], tmp_path)
    if frames == 300:
        assert _sha1(ref) == "02de34fb5258838ab763673c00f5c8918d27751b"
    bs, _ = oh.encode_sequence(yuv, 1280, 720, lib_path=lib, iDLayerQp=24, uiIntraPeriod=1, fMaxFrameRate=30.0, iTargetBitrate=5000000)
    assert bs == ref


def _run_qcif(lib, ref_tools, tmp_path):
    yuv = _decode(ref_tools, "BA_MW_D.264", tmp_path)[: 176 * 144 * 3 // 2 * 60]
    ref = _ref_encode(ref_tools, yuv, 176, 144, ["-iper", "30", "-qp", "30", "-complexity", "1"], tmp_path)
    bs, _ = oh.encode_sequence(yuv, 176, 144, lib_path=lib, iDLayerQp=30, uiIntraPeriod=30, fMaxFrameRate=30.0,
                               iTargetBitrate=5000000, iComplexityMode=1)
    assert bs == ref


def _run_config1(lib, ref_tools, tmp_path):
    """BASELINE config 1 at constant QP: CiscoVT2people_160x96, 1 slice, QP 24 (h264enc welsenc.cfg -rc -1 -lqp 0 24 gives
    dd643761...: two temporal layers there; the single-layer stream is compared with the oracle run live)."""
    yuv = open(os.path.join(RES, "CiscoVT2people_160x96_6fps.yuv"), "rb").read()
    ref = _ref_encode(ref_tools, yuv, 160, 96, ["-iper", "0", "-qp", "24", "-scene", "1"], tmp_path)
    bs, _ = oh.encode_sequence(yuv, 160, 96, lib_path=lib, iDLayerQp=24, uiIntraPeriod=0, fMaxFrameRate=30.0, iTargetBitrate=5000000,
                               bEnableSceneChangeDetect=1)
    assert bs == ref


def test_1080p_p_frames_4_slices_survey_hash(emu_lib, ref_tools, tmp_path):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    _run_1080p(emu_lib, ref_tools, tmp_path)


def test_720p_intra_survey_hash_prefix(emu_lib, ref_tools, tmp_path):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    _run_720p_intra(emu_lib, ref_tools, tmp_path, 24)


def test_qcif_clip_medium_complexity(emu_lib, ref_tools, tmp_path):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    _run_qcif(emu_lib, ref_tools, tmp_path)


def test_config1_clip(emu_lib, ref_tools, tmp_path):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    _run_config1(emu_lib, ref_tools, tmp_path)


@pytest.mark.gpu
def test_hip_1080p_p_frames_4_slices_survey_hash(hip_lib, ref_tools, tmp_path):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    _run_1080p(hip_lib, ref_tools, tmp_path)


@pytest.mark.gpu
def test_hip_720p_intra_300_frames_survey_hash(hip_lib, ref_tools, tmp_path):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    _run_720p_intra(hip_lib, ref_tools, tmp_path, 300)


@pytest.mark.gpu
def test_hip_qcif_clip_medium_complexity(hip_lib, ref_tools, tmp_path):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    _run_qcif(hip_lib, ref_tools, tmp_path)


@pytest.mark.gpu
def test_hip_config1_clip(hip_lib, ref_tools, tmp_path):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    _run_config1(hip_lib, ref_tools, tmp_path)
