"""The diagnostic tools must keep working: the CAVLC syntax dumper parses what the encoder writes."""
import io
import os
import sys

import openh264_amd as oh
from openh264_amd.utils.synth import synth_sequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_h264_parse_reads_our_streams(emu_lib):
    import h264_parse
    w, h, n = 176, 144, 4
    bs, _ = oh.encode_sequence(synth_sequence(w, h, n), w, h, lib_path=emu_lib, iDLayerQp=26, uiIntraPeriod=0, fMaxFrameRate=30.0,
                               iTargetBitrate=5000000, uiSliceMode=1, uiSliceNum=3, iComplexityMode=1)
    out = io.StringIO()
    h264_parse.parse_stream(bs, out=out, levels=True)
    text = out.getvalue()
    assert text.count("SLICE pic") == 3 * n
    mbs = [ln for ln in text.splitlines() if ln.startswith("  mb ")]
    assert len(mbs) == 99 * n                     # every macroblock of every picture was parsed (skips included)
    assert any("P16x16" in ln or "P_Skip" in ln for ln in mbs) and any("I4x4" in ln or "I16x16" in ln for ln in mbs)
