"""Parity on the reference's own test clips (res/*.264 decoded with the reference decoder), incl. the
1080p / 4-slice / QP 24 P-frame stream whose SHA1 the survey recorded from the stock h264enc
(bb6dba56327985bc94c60ca1e7a004b4fb543513).  Needs /root/reference (build container only)."""
import hashlib
import os
import subprocess

import pytest

import openh264_amd as oh

RES = "/root/reference/res"
pytestmark = pytest.mark.skipif(not os.path.isdir(RES), reason="reference tree not present")


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


def test_1080p_p_frames_4_slices_survey_hash(emu_lib, ref_tools, tmp_path):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    yuv = _decode(ref_tools, "VID_1920x1080_cavlc_temporal_direct.264", tmp_path)
    assert len(yuv) == 167961600                                   # 54 frames
    ref = _ref_encode(ref_tools, yuv, 1920, 1080, ["-iper", "0", "-qp", "24", "-slcmd", "1", "-slcnum", "4"], tmp_path)
    assert hashlib.sha1(ref).hexdigest() == "bb6dba56327985bc94c60ca1e7a004b4fb543513"
    bs, _ = oh.encode_sequence(yuv, 1920, 1080, lib_path=emu_lib, iDLayerQp=24, uiIntraPeriod=0, fMaxFrameRate=30.0,
                               iTargetBitrate=5000000, uiSliceMode=1, uiSliceNum=4)
    assert bs == ref


def test_720p_intra_survey_hash_prefix(emu_lib, ref_tools, tmp_path):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    yuv = _decode(ref_tools, "VID_1280x720_cavlc_temporal_direct.264", tmp_path)[: 1280 * 720 * 3 // 2 * 24]
    ref = _ref_encode(ref_tools, yuv, 1280, 720, ["-iper", "1", "-qp", "24"], tmp_path)
    bs, _ = oh.encode_sequence(yuv, 1280, 720, lib_path=emu_lib, iDLayerQp=24, uiIntraPeriod=1, fMaxFrameRate=30.0, iTargetBitrate=5000000)
    assert bs == ref


def test_qcif_clip_medium_complexity(emu_lib, ref_tools, tmp_path):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    yuv = _decode(ref_tools, "BA_MW_D.264", tmp_path)[: 176 * 144 * 3 // 2 * 60]
    ref = _ref_encode(ref_tools, yuv, 176, 144, ["-iper", "30", "-qp", "30", "-complexity", "1"], tmp_path)
    bs, _ = oh.encode_sequence(yuv, 176, 144, lib_path=emu_lib, iDLayerQp=30, uiIntraPeriod=30, fMaxFrameRate=30.0,
                               iTargetBitrate=5000000, iComplexityMode=1)
    assert bs == ref
