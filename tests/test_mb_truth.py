"""The per-macroblock ground-truth trace (tools/mb_truth.py, welship_hooks.cpp TraceWriteMbSyn).

Both paths hand every macroblock to the entropy writer through SWelsFuncPtrList::pfWelsSpatialWriteMbSyn; with WELS_HIP_MB_TRACE set the hooked
reference wraps that slot and writes what the writer is about to code.  Here: the reference's C path and the path through the hooks leave the same
lines (types, cbp, QP, vector differences, total_coeff, intra modes, level hashes) for camera pictures with two slices, for a CABAC session and for
size-limited slices (macroblocks coded twice), and two runs that DO differ (another QP) are told apart macroblock by macroblock."""
import io
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ref_enc_hip")) and not os.environ.get("WELSHIP_REQUIRE_ORACLE"),
                                reason="oracle/_ref (hooked reference) not built")

CASES = [
    ("two_slices", ["-rc", "-1", "-qp", "26", "-slcmd", "1", "-slcnum", "2"]),
    ("cabac_rate_control", ["-rc", "1", "-bitrate", "300000", "-cabac", "1"]),
    ("size_limited_slices", ["-rc", "-1", "-qp", "24", "-slcmd", "3", "-slcsize", "600"]),
]


def _clip(tmp_path, frames=6):
    from openh264_amd.utils.synth import synth_sequence
    src = str(tmp_path / "in.yuv")
    open(src, "wb").write(synth_sequence(320, 192, frames))
    return ["-i", src, "-w", "320", "-h", "192"]


def _both(lib, tmp_path, flags):
    import mb_truth
    work = str(tmp_path)
    mb_truth.run(flags, work, "c_path", {"WELS_HIP": "0"})
    err = mb_truth.run(flags, work, "tested", {"WELSHIP_LIB": lib})
    assert err.count("welship hooks: did") >= 5, err[-1500:]
    assert open(os.path.join(work, "c_path.264"), "rb").read() == open(os.path.join(work, "tested.264"), "rb").read()
    out = io.StringIO()
    n = mb_truth.compare(os.path.join(work, "c_path.mbs"), os.path.join(work, "tested.mbs"), out=out)
    mbs, order = mb_truth.load(os.path.join(work, "c_path.mbs"))
    return n, out.getvalue(), mbs, order


@pytest.mark.parametrize("name,flags", CASES, ids=[c[0] for c in CASES])
def test_trace_of_the_c_path_equals_the_trace_through_the_hooks(emu_lib, tmp_path, name, flags):
    n, text, mbs, order = _both(emu_lib, tmp_path, _clip(tmp_path) + flags)
    assert n == 0, text
    assert len(order) == 6 * 240, len(order)                          # every macroblock of every picture, once
    kinds = {m.split()[8] for m in mbs.values()}
    assert {"skip", "p16x16"} <= kinds and ({"i4x4", "i16x16"} & kinds), kinds


def test_two_runs_that_differ_are_told_apart(emu_lib, tmp_path):
    import mb_truth
    base = _clip(tmp_path, 3) + ["-rc", "-1"]
    mb_truth.run(base + ["-qp", "26"], str(tmp_path), "a", {"WELSHIP_LIB": emu_lib})
    mb_truth.run(base + ["-qp", "27"], str(tmp_path), "b", {"WELSHIP_LIB": emu_lib})
    out = io.StringIO()
    n = mb_truth.compare(str(tmp_path / "a.mbs"), str(tmp_path / "b.mbs"), out=out)
    assert n > 100 and "picture 0 layer 0 macroblock 0" in out.getvalue() and "qp 26" in out.getvalue() and "qp 27" in out.getvalue(), out.getvalue()[:2000]


@pytest.mark.gpu
def test_trace_through_the_hooks_on_the_mi355x(hip_lib, tmp_path):
    n, text, mbs, order = _both(hip_lib, tmp_path, _clip(tmp_path) + CASES[0][1])
    assert n == 0, text
