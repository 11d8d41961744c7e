"""Parity spot check of the PLAIN variant of the P kernel on the device: session groups through a library built with -DWH_PLAIN_KERNEL=1
against the single-session encoder of the default library (which never takes the variant).  python tools/plain_check.py <plain lib>"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import openh264_amd as oh
import test_multi_rank as T
from openh264_amd.parallel import encode_sessions_sharded

lib = os.path.abspath(sys.argv[1])
base = os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else oh.DEFAULT_LIB
inputs = T._inputs()
digs = encode_sessions_sharded(T._make_group_factory(lib), inputs, T.FRAMES)
for s in range(T.SESSIONS):
    bs, _ = oh.encode_sequence(b"".join(inputs[s]), T.W, T.H, lib_path=base, iDLayerQp=26, uiIntraPeriod=0, fMaxFrameRate=30.0, iTargetBitrate=500000)
    assert hashlib.sha1(bs).hexdigest() == digs[s], s
seqs, got = T._pipelined_vs_synchronous(lib, 320, 192, 8, 24, ("synth", "checker5", "synth", "pan7"), 3, intra_period=5, threads=4, ahead=2)
for s, yuv in enumerate(seqs):
    bs, _ = oh.encode_sequence(yuv, 320, 192, lib_path=base, iDLayerQp=24, uiIntraPeriod=5, fMaxFrameRate=30.0, iTargetBitrate=5000000, bEnableSceneChangeDetect=False)
    assert bs == got[s], s
print("plain variant: %d + %d streams identical to the single-session encoder of the default library" % (T.SESSIONS, len(seqs)))
