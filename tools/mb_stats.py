"""What the macroblocks of a 1080p P picture become and what their searches cost, counted by the CPU test build (WELSHIP_WIN_STATS=1).
usage: mb_stats.py [synthetic|res] [frames]   -- bench.py's synthetic content or the reference's own 1080p clip (oracle/_ref/res)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WELSHIP_WIN_STATS"] = "1"
from openh264_amd import build as B
import openh264_amd as oh
from openh264_amd.utils.synth import synth_sequence
lib = B.build_emu()
w, h = 1920, 1080
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
if len(sys.argv) > 1 and sys.argv[1] == "res":
    import bench
    yuv = bench.decode_res_clip("VID_1920x1080_cavlc_temporal_direct.264")[: n * w * h * 3 // 2]
else:
    yuv = synth_sequence(w, h, n)
bs, _ = oh.encode_sequence(yuv, w, h, lib_path=lib, iDLayerQp=24, uiIntraPeriod=0, fMaxFrameRate=30.0, iTargetBitrate=5000000, uiSliceMode=1, uiSliceNum=4)
print("frames", n, "bytes", len(bs))
