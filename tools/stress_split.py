"""P pictures full of intra macroblocks through k_inter_split (a launch of few slices: every slice on several compute units), against the CPU test
build of the same kernel sources: noise that changes completely from picture to picture at a low QP, so that the macroblock types at one position
alternate between Intra4x4 (all nine modes) and everything else -- what makes a neighbour state read from a stale cache line visible.
   python tools/stress_split.py <iterations> [<lib>] [<w> <h> <frames> <qp>]
Round 6: written after one P picture of one session differed in one of six runs of the GPU tier (the Intra4x4 neighbour-mode cache read the neighbour
states with plain loads, tests/test_multi_rank.py::test_hip_pipelined_group_reencodes_after_cavlc_overflow)."""
import sys, os, time, hashlib
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import openh264_amd as oh
from openh264_amd import build as B

n = int(sys.argv[1])
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "openh264_amd", "libwelship.so")
w, h, frames, qp = (int(x) for x in sys.argv[3:7]) if len(sys.argv) > 6 else (320, 192, 6, 24)
emu = B.build_emu()
bad = 0
t0 = time.time()
for i in range(n):
    rng = np.random.default_rng(1000 + i)
    # blocks of noise of varying contrast: flat areas (Intra16x16 / inter), busy areas (Intra4x4), different in every picture
    yuv = bytearray()
    for f in range(frames):
        amp = rng.integers(0, 120, size=((h + 15) // 16, (w + 15) // 16)).repeat(16, 0).repeat(16, 1)[:h, :w]
        y = np.clip(128 + (rng.standard_normal((h, w)) * amp), 0, 255).astype(np.uint8)
        c = rng.integers(100, 156, size=(h // 2, w), dtype=np.uint8)
        yuv += y.tobytes() + c.tobytes()
    yuv = bytes(yuv)
    kw = dict(iDLayerQp=qp, uiIntraPeriod=0, fMaxFrameRate=30.0, iTargetBitrate=5000000, bEnableSceneChangeDetect=False)
    want, _ = oh.encode_sequence(yuv, w, h, lib_path=emu, **kw)
    got, _ = oh.encode_sequence(yuv, w, h, lib_path=lib, **kw)
    if got != want:
        bad += 1
        first = next((k for k in range(min(len(got), len(want))) if got[k] != want[k]), min(len(got), len(want)))
        print("iteration", i, "differs from byte", first, "of", len(want), "(access unit %d)" % want[:first].count(b"\x00\x00\x00\x01"), flush=True)
print("iterations", n, "bad", bad, os.path.basename(lib), "WELSHIP_MD_SPLIT", os.environ.get("WELSHIP_MD_SPLIT"), "%dx%d x %d qp %d" % (w, h, frames, qp), "%.1f s" % (time.time() - t0))
