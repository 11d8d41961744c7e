"""Size-limited slices (SM_SIZELIMITED_SLICE) through the dispatch-table binding (default on; WELS_HIP_DYNSLICE=0 declines them).

Where such a slice ends is decided by the entropy writer, macroblock by macroblock (DynSlcJudgeSliceBoundaryStepBack,
codec/encoder/core/src/svc_encode_slice.cpp:1741-1790): when a macroblock would make the slice larger than its limit, the writer
takes it back, ends the slice before it, and the macroblock is DECIDED AGAIN as the first one of the next slice -- without its
neighbours (WelsMdInterMbLoopOverDynamicSlice :1901-2010, WelsISliceMdEncDynamic :601-680).  The binding lets the device code ahead
of the writer (one and a half slices' worth of macroblocks per call, as if the slice never ended) and repeats the call from the
macroblock a new slice begins with (WelsHipFrameJob::iDynSlice, include/welship.h; integration/welship_hooks.cpp HipCodeSlice).

Checked here on the CPU test build of the kernels (tests/emu): random sessions byte for byte against the unmodified reference
(tools/fuzz_dynslice.py: 421 .. 3000 bytes per slice, i.e. from slices shorter than a macroblock row -- dozens per picture -- to one
slice per picture; all rate-control modes, temporal layers, LTR, denoising, background / scene-change detection, the three
deblocking modes, I pictures in mid-stream, CAVLC and CABAC, one to four slice threads, camera video and screen content).  The reference table's own size-limited rows: tests/test_hooks_sha1.py.
The comparison (tools/fuzz_dynslice.py one_case): the unmodified reference runs FIRST, several times when the session has slice threads,
and the hooked run's stream must be one of the streams it produced; every result line names the set ("reference: 1 stream in 3 runs
{ef6f29ba}; with the hooks: ef6f29ba").  One class of sessions has a set larger than one -- slice threads with screen content or rate
control: the reference's own output depends on which pool thread runs which slice task (root cause in one_case; measured in
profiles/r04_size_limited_slices_screen_threads_reference_varies.txt: seed 21001, the case the round-3 driver run reported as DIFF, is
such a session -- 4 streams in 60 runs of the unmodified reference, the hooked run's stream one of them).  Every other session must
match the reference's ONE stream.
"""
import os
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ref_enc_hip")), reason="oracle/_ref (hooked reference) not built")


def _fuzz(lib, seeds):
    import fuzz_dynslice
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory() as tmp:
        with ThreadPoolExecutor(8) as ex:
            res = list(ex.map(lambda s: fuzz_dynslice.one_case(s, lib, tmp), seeds))
    bad = [(s, m) for s, m, ok in res if not ok]
    assert not bad, bad[0]
    ran = [m for _, m, _ in res if m.startswith("ok")]
    assert len(ran) >= len(seeds) * 3 // 4
    # the cases really were sessions of many slices: more slices than pictures, and more device calls than pictures
    slices = sum(int(m.split("device,")[1].split()[0]) for m in ran)
    pictures = sum(int(m.split("frames,")[1].split()[0]) for m in ran)
    assert slices > 4 * pictures


def test_random_sessions_on_emulation(emu_lib):
    _fuzz(emu_lib, range(1000, 1016))


def test_random_sessions_on_emulation_reverse_lane_order():
    """The same through the test build that walks the lanes of a lane block from 63 down (tests/test_frame_parity.py)."""
    from openh264_amd import build as B
    _fuzz(B.build_emu(defines=("WH_EMU_REVERSE",), tag="wh_emu_reverse"), range(7000, 7008))


def test_reencoding_after_cavlc_overflow_inside_size_limited_slices(emu_lib):
    """TRY_REENCODING (svc_encode_slice.cpp:564-576,1845-1867) under size-limited slices: constant QP 0..12 on saturated content makes CAVLC
    levels overflow; the writer takes the macroblock back and the device codes it -- and what it had coded ahead of it -- again at QP + 2
    (an MB range repeated with WelsHipFrameJob::pReencode; the QP_Y chain for the filter follows the slices where the states' slice
    index changes).  One to four slice threads."""
    import fuzz_dynslice
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory() as tmp:
        with ThreadPoolExecutor(3) as ex:
            res = list(ex.map(lambda s: fuzz_dynslice.one_case(s, emu_lib, tmp, True, 4, True), range(11012, 11024)))
    bad = [(s, m) for s, m, ok in res if not ok]
    assert not bad, bad[0]
    assert sum(1 for _, m, _ in res if "repeated after a CAVLC overflow" in m) >= 2


def test_installer_default_and_switch(emu_lib, tmp_path):
    """On by default since the path has run on the MI355X (profiles/r03_size_limited_slices_*_mi355x.txt); WELS_HIP_DYNSLICE=0 leaves
    the session to the reference's C path and the installer says why."""
    import subprocess
    from openh264_amd.utils.synth import synth_sequence
    src = str(tmp_path / "c.yuv")
    open(src, "wb").write(synth_sequence(176, 144, 3))
    env = dict(os.environ, WELSHIP_LIB=emu_lib, WELS_HIP_TRACE="1")
    cmd = [os.path.join(REF, "ref_enc_hip"), "-i", src, "-w", "176", "-h", "144", "-o", str(tmp_path / "o.264"), "-quiet", "-slcmd", "3",
           "-slcsize", "600", "-threads", "1", "-rc", "-1", "-qp", "26"]
    env.pop("WELS_HIP_DYNSLICE", None)
    p = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0 and "welship hooks: installed" in err and "picture complete" in err
    env["WELS_HIP_DYNSLICE"] = "0"
    p = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0 and "not installed" in err and "WELS_HIP_DYNSLICE" in err
    # ... and a simulcast session with size-limited slices stays on the C path
    env["WELS_HIP_DYNSLICE"] = "1"
    p = subprocess.run([os.path.join(REF, "ref_enc_hip"), "-i", src, "-w", "176", "-h", "144", "-o", str(tmp_path / "o.264"), "-quiet", "-slcmd", "3",
                        "-slcsize", "600", "-threads", "1", "-rc", "-1", "-qp", "26", "-simulcast", "96", "80"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0 and "not installed" in err and "one spatial layer" in err


def test_random_screen_content_sessions_on_emulation(emu_lib):
    """Screen content (-usage 1) under size-limited slices: the chain of the 8x8 searches' costs is kept per macroblock then
    (WhSccJob::chain_mb: a slice can begin, or a range be coded again, at any macroblock) and the feature search's cost-down sums travel
    in the records (WhMbRecord::fme_down: the host adds up the macroblocks the writer really took)."""
    import fuzz_dynslice
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory() as tmp:
        with ThreadPoolExecutor(3) as ex:      # (few at a time: with slice threads the reference's own output can depend on the machine's load, see tools/fuzz_dynslice.py)
            res = list(ex.map(lambda s: fuzz_dynslice.one_case(s, emu_lib, tmp, True, 4, False, True), range(21000, 21016)))
    bad = [(s, m) for s, m, ok in res if not ok]
    assert not bad, bad[0]
    assert sum(1 for _, m, _ in res if m.startswith("ok")) >= 12


def test_random_sessions_with_slice_threads_on_emulation(emu_lib):
    """Several slice threads: the picture is split into one partition per thread, every partition is sliced on its own by its thread's
    task (CWelsConstrainedSizeSlicingEncodingTask, wels_task_encoder.cpp:231-325) and the tasks call the device concurrently."""
    import fuzz_dynslice
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory() as tmp:
        with ThreadPoolExecutor(3) as ex:
            res = list(ex.map(lambda s: fuzz_dynslice.one_case(s, emu_lib, tmp, True, 4), range(5000, 5012)))
    bad = [(s, m) for s, m, ok in res if not ok]
    assert not bad, bad[0]
    assert sum(1 for _, m, _ in res if m.startswith("ok") and "-threads 1 " not in m) >= 6


def test_harness_catches_one_corrupted_macroblock_in_one_slice_threads_range(emu_lib):
    """The comparison is only worth something if it fails when it should: the test build is told to change ONE coefficient level of ONE
    macroblock's record whenever a ranged call codes it (tests/emu/emu_backend.cpp, WELSHIP_EMU_CORRUPT_MB) -- in the last slice thread's
    partition of a constant-QP session with three slice threads.  The unmodified reference gives one stream for such a session, so
    nothing but an exact match passes; the case must come back as DIFF (or as a failed run: the writer's own checks may notice first)."""
    import fuzz_dynslice
    with tempfile.TemporaryDirectory() as tmp:
        force = {"-threads": "3", "-qp": "10", "-rc": "-1", "-cabac": "0"}
        seed, msg, ok = fuzz_dynslice.one_case(1005, emu_lib, tmp, True, 4, False, False, force=force)
        assert ok and msg.startswith("ok") and "reference: 1 stream in 3 runs" in msg, msg
        w, h = (int(t) for t in msg.split()[1].split("x"))
        num_mb = ((w + 15) // 16) * ((h + 15) // 16)
        seed, msg, ok = fuzz_dynslice.one_case(1005, emu_lib, tmp, True, 4, False, False, force=force, env={"WELSHIP_EMU_CORRUPT_MB": str(num_mb - 3)})
        assert not ok and (msg.startswith("DIFF") or msg.startswith("FAILED")), msg


# ---- GPU tier (first run on the MI355X: round 3, profiles/r03_size_limited_slices_*_mi355x.txt)
@pytest.mark.gpu
def test_random_sessions_on_the_mi355x(hip_lib):
    _fuzz(hip_lib, range(1000, 1016))


@pytest.mark.gpu
def test_random_screen_content_and_low_qp_sessions_on_the_mi355x(hip_lib):
    import fuzz_dynslice
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory() as tmp:
        with ThreadPoolExecutor(4) as ex:
            res = list(ex.map(lambda s: fuzz_dynslice.one_case(s, hip_lib, tmp, True, 4, False, True), range(21000, 21012)))
            res += list(ex.map(lambda s: fuzz_dynslice.one_case(s, hip_lib, tmp, True, 4, True), range(11012, 11020)))
    bad = [(s, m) for s, m, ok in res if not ok]
    assert not bad, bad[0]
