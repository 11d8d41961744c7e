#!/usr/bin/env python3
"""Generate tests/golden/golden.json: bitstream SHA1s produced by the *reference* (oracle/_ref,
built from /root/reference by oracle/Makefile) on the deterministic synthetic sequences of
openh264_amd/utils/synth.py.  Run in the build container (the reference does not travel):

    python tools/make_golden.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openh264_amd.utils.synth import synth_sequence  # noqa: E402

# name -> (w, h, frames, ref_enc flags, WelsHipEncParam overrides)
CASES = {
    "smoke_160x96_iper1_qp24": (160, 96, 3, ["-iper", "1", "-qp", "24"], dict(uiIntraPeriod=1, iDLayerQp=24)),
    "i_176x144_qp24": (176, 144, 4, ["-iper", "1", "-qp", "24"], dict(uiIntraPeriod=1, iDLayerQp=24)),
    "i_176x144_qp36_c1": (176, 144, 3, ["-iper", "1", "-qp", "36", "-complexity", "1"], dict(uiIntraPeriod=1, iDLayerQp=36, iComplexityMode=1)),
    "i_320x192_qp10": (320, 192, 2, ["-iper", "1", "-qp", "10"], dict(uiIntraPeriod=1, iDLayerQp=10)),
    "i_320x192_qp30_idc1": (320, 192, 2, ["-iper", "1", "-qp", "30", "-deblock", "1"], dict(uiIntraPeriod=1, iDLayerQp=30, iLoopFilterDisableIdc=1)),
    "i_152x100_qp24_crop": (152, 100, 3, ["-iper", "1", "-qp", "24"], dict(uiIntraPeriod=1, iDLayerQp=24)),
    "i_1280x720_qp24": (1280, 720, 2, ["-iper", "1", "-qp", "24"], dict(uiIntraPeriod=1, iDLayerQp=24)),
    "i_1920x1080_qp24": (1920, 1080, 1, ["-iper", "1", "-qp", "24"], dict(uiIntraPeriod=1, iDLayerQp=24)),
    "i_640x368_qp24_constid": (640, 368, 3, ["-iper", "1", "-qp", "24", "-spsid", "0"], dict(uiIntraPeriod=1, iDLayerQp=24, eSpsPpsIdStrategy=0)),
    # P pictures (IDR + P...), diamond ME + fractional refinement + skip/intra/inter decisions
    "p_176x144_qp24": (176, 144, 8, ["-iper", "0", "-qp", "24"], dict(uiIntraPeriod=0, iDLayerQp=24)),
    "p_320x192_qp28_c1": (320, 192, 6, ["-iper", "0", "-qp", "28", "-complexity", "1"], dict(uiIntraPeriod=0, iDLayerQp=28, iComplexityMode=1)),
    "p_320x192_qp20_c2_iper4": (320, 192, 9, ["-iper", "4", "-qp", "20", "-complexity", "2"], dict(uiIntraPeriod=4, iDLayerQp=20, iComplexityMode=2)),
    "p_640x368_qp24_4slices": (640, 368, 5, ["-iper", "0", "-qp", "24", "-slcmd", "1", "-slcnum", "4"], dict(uiIntraPeriod=0, iDLayerQp=24, uiSliceMode=1, uiSliceNum=4)),
    "p_640x368_qp32_3slices_idc2": (640, 368, 4, ["-iper", "0", "-qp", "32", "-slcmd", "1", "-slcnum", "3", "-deblock", "2"], dict(uiIntraPeriod=0, iDLayerQp=32, uiSliceMode=1, uiSliceNum=3, iLoopFilterDisableIdc=2)),
    "p_152x100_qp24_crop": (152, 100, 6, ["-iper", "0", "-qp", "24"], dict(uiIntraPeriod=0, iDLayerQp=24)),
    "p_1280x720_qp24": (1280, 720, 3, ["-iper", "0", "-qp", "24"], dict(uiIntraPeriod=0, iDLayerQp=24)),
    "p_1920x1080_qp24_4slices": (1920, 1080, 3, ["-iper", "0", "-qp", "24", "-slcmd", "1", "-slcnum", "4"], dict(uiIntraPeriod=0, iDLayerQp=24, uiSliceMode=1, uiSliceNum=4)),
}
COMMON = ["-rc", "-1", "-fps", "30", "-quiet"]


def main():
    enc = os.path.join(ROOT, "oracle", "_ref", "ref_enc")
    dec = os.path.join(ROOT, "oracle", "_ref", "ref_dec")
    out = {}
    for name, (w, h, n, flags, params) in CASES.items():
        yuv = synth_sequence(w, h, n)
        with tempfile.TemporaryDirectory() as td:
            fi, fo, fd = os.path.join(td, "in.yuv"), os.path.join(td, "o.264"), os.path.join(td, "d.yuv")
            open(fi, "wb").write(yuv)
            subprocess.check_call([enc, "-i", fi, "-w", str(w), "-h", str(h), "-o", fo] + flags + COMMON, stdout=subprocess.DEVNULL)
            bs = open(fo, "rb").read()
            subprocess.check_call([dec, fo, fd], stdout=subprocess.DEVNULL)
            rec = open(fd, "rb").read()
        out[name] = {"w": w, "h": h, "frames": n, "ref_flags": flags + COMMON, "params": params,
                     "input_sha1": hashlib.sha1(yuv).hexdigest(), "bytes": len(bs), "sha1": hashlib.sha1(bs).hexdigest(),
                     "recon_sha1": hashlib.sha1(rec).hexdigest()}
        print(name, len(bs), out[name]["sha1"])
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "golden.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
