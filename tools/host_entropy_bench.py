#!/usr/bin/env python3
"""Host side of a session group's frame step, timed on THIS machine's cores: WelsHipGroupFinish = expansion of the packed
records + CAVLC + NAL packing of every session's picture (csrc/host/encoder.cpp, entropy_cavlc.cpp, bitwriter.h).

The macroblock records come from the CPU test build of the kernels (tests/emu), so no GPU is needed: the host code that is
timed is the same object code libwelship.so contains.  One thread; the figure is ms per 1080p picture and thread, the term
that bounds the end-to-end legs of bench.py (PCIe-inclusive rate).

  python tools/host_entropy_bench.py [--frames 6] [--sessions 2] [--clip]      (--clip: the reference's 1080p clip)
The SHA1 of all bitstreams is printed so that two builds can be compared byte for byte."""
import argparse
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--sessions", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--qp", type=int, default=24)
    ap.add_argument("--clip", action="store_true")
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    from openh264_amd import build as B
    lib = a.lib or B.build_emu()
    import openh264_amd as oh
    from openh264_amd.utils.synth import synth_sequence
    import ctypes as C
    w, h = a.width, a.height
    fsz = w * h * 3 // 2
    if a.clip:
        import bench
        yuv = bench.decode_res_clip("VID_1920x1080_cavlc_temporal_direct.264")
        assert yuv is not None and (w, h) == (1920, 1080)
    else:
        yuv = synth_sequence(w, h, a.frames + a.sessions * 3)
    e = oh.Encoder(lib)
    p = e.GetDefaultParams()
    e.close()
    p.iPicWidth, p.iPicHeight, p.iDLayerQp, p.fMaxFrameRate, p.iTargetBitrate = w, h, a.qp, 30.0, 5000000
    p.uiIntraPeriod = 0
    p.uiSliceMode, p.uiSliceNum = 1, 4
    g = oh.EncoderGroup(p, a.sessions, ring_slots=2, host_threads=1, lib_path=lib)
    L = g._lib
    sha = hashlib.sha1()
    t_fin = t_dev = 0.0
    nbytes = 0
    for f in range(a.frames):
        for s in range(a.sessions):
            k = (s * 3 + f) % (len(yuv) // fsz)
            g.upload(s, f % 2, yuv[k * fsz:(k + 1) * fsz])
        t0 = time.perf_counter()
        rc = L.WelsHipGroupBegin(g._h, f % 2) or L.WelsHipGroupRunDevice(g._h, 1)
        assert rc == 0, rc
        t1 = time.perf_counter()
        infos = (oh.SFrameBSInfo * g.n)()
        rc = L.WelsHipGroupFinish(g._h, infos)
        t2 = time.perf_counter()
        assert rc == 0, rc
        for info in infos:
            for li in range(info.iLayerNum):
                Ly = info.sLayerInfo[li]
                b = C.string_at(Ly.pBsBuf, sum(Ly.pNalLengthInByte[k] for k in range(Ly.iNalCount)))
                sha.update(b)
                if f > 0:
                    nbytes += len(b)
        if f > 0:                      # P pictures only
            t_fin += t2 - t1
            t_dev += t1 - t0
    n = (a.frames - 1) * a.sessions
    print("host finish: %.3f ms per P picture and thread (%d pictures of %dx%d, %.0f bytes each; emulated device passes %.0f ms each)"
          % (1e3 * t_fin / n, n, w, h, nbytes / n, 1e3 * t_dev / n))
    print("WelsHipGroupHostStats (IDR included):", g.host_stats())
    print("sha1 of all bitstreams:", sha.hexdigest())


if __name__ == "__main__":
    main()
