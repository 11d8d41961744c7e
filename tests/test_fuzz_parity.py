"""Randomised bitstream parity against oracle/_ref run live (tools/fuzz_parity.py): random picture sizes, QPs (0..51),
slice counts, complexity modes, deblocking modes/offsets, cropping, parameter-set id strategy, forced IDRs and six
content classes, including streams that hit the reference's CAVLC-overflow re-encode loop and its buffer limits."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, ref_tools, seed=7, env=None):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--cases", "24", "--seed", str(seed)] + extra,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500, env=dict(os.environ, **(env or {})))
    out = r.stdout.decode()
    assert r.returncode == 0, out[-4000:]
    assert "24 cases, 0 failed" in out


# single cases that once differed, kept as regressions: (seed, case, what it was)
REGRESSIONS = [
    (176, 393, [], "level 1 stream: integer search range 63 (GetMvMvdRange), MV at the range edge"),
    (9203, 3525, ["--level1", "--max-mbs", "99"], "level 1b Baseline stream is looked up as level 1.1: search range 64"),
    (611, 9, ["--options"], "IDR interval 1 -> 2 in mid-stream: the last all-IDR picture becomes a reference (border expansion)"),
]


def _run_one(seed, case, extra, ref_tools):
    extra = list(extra)
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "--seed", str(seed), "--only", str(case)] + extra,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "1 cases, 0 failed" in out, out[-4000:]


@pytest.mark.parametrize("seed,case,flags,what", REGRESSIONS)
def test_fuzz_regression_emu(seed, case, flags, what, emu_lib, ref_tools):
    _run_one(seed, case, flags, ref_tools)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,case,flags,what", REGRESSIONS)
def test_fuzz_regression_hip(seed, case, flags, what, hip_lib, ref_tools):
    _run_one(seed, case, flags + ["--hip"], ref_tools)


def test_fuzz_emu(emu_lib, ref_tools):
    _run([], ref_tools)


@pytest.mark.gpu
def test_fuzz_hip(hip_lib, ref_tools):
    _run(["--hip"], ref_tools)


# More draws of the same generator: the search windows of a wave are small (kernels/inter_mb.h: 38 rows of luma, 24 of chroma) and follow
# the search as it walks (wh_win_need, the diamond's step budget), so the fast-motion and noise classes of other seeds walk other paths
# through the reloads.  (Rounds 3-4 had a second scheduler here -- runs of macroblocks with sliding windows, k_inter_rows -- which was
# measured slower at every run length, profiles/r03_run_length_sweep.txt, and was removed in round 5.)
@pytest.mark.parametrize("seed", [31, 37, 41])
def test_fuzz_emu_more_seeds(emu_lib, ref_tools, seed):
    _run([], ref_tools, seed=seed)


def test_fuzz_emu_tiny_windows(ref_tools):
    """The same generator on the test build whose search windows hold little more than the block (tests/test_frame_parity.py
    test_emu_tiny_search_windows): every reload path of kernels/inter_mb.h, on random sizes and the fast-motion / noise classes."""
    from openh264_amd import build as B
    lib = B.build_emu(defines=("WH_WIN_ROWS=30", "WH_WIN_MARGIN_Y=7", "WH_WIN_START=1", "WH_CWIN_ROWS=12"), tag="tiny_windows")
    _run(["--lib", lib], ref_tools, seed=43)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [23, 29])
def test_fuzz_hip_more_seeds(hip_lib, ref_tools, seed):
    _run(["--hip"], ref_tools, seed=seed)
