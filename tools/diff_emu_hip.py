"""Debug aid (GPU box): encode the same synthetic sequence with the CPU wave-emulation test build and with the HIP
library, frame by frame; report the first frame whose bitstream or reconstruction differs and the MBs involved.
usage: python tools/diff_emu_hip.py W H FRAMES [key=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openh264_amd as oh
from openh264_amd import build as B
from openh264_amd.utils.synth import synth_sequence

w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
params = dict(fMaxFrameRate=30.0, iTargetBitrate=5000000, iDLayerQp=24, uiIntraPeriod=0)
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    params[k] = int(v)
yuv = synth_sequence(w, h, n)
fsz = w * h * 3 // 2
encs = []
for lib in (os.path.join(os.path.dirname(B.__file__), "..", "tests", "emu", "libwelship_emu.so"), B.build_hip()):
    if not os.path.exists(lib):
        lib = B.build_emu()
    e = oh.Encoder(lib)
    p = e.GetDefaultParams()
    p.iPicWidth, p.iPicHeight = w, h
    for k, v in params.items():
        setattr(p, k, v)
    assert e.InitializeExt(p) == 0, e.last_error()
    encs.append(e)
for i in range(n):
    outs = []
    for e in encs:
        rc, _, bs, _ = e.EncodeFrame(yuv[i * fsz:(i + 1) * fsz], timestamp=i * 33)
        outs.append((bytes(bs), e.GetReconFrame()))
    (b0, r0), (b1, r1) = outs
    if b0 != b1 or r0 != r1:
        print("frame %d differs: bitstream %s (len %d vs %d), recon %s" % (i, b0 != b1, len(b0), len(b1), r0 != r1))
        bad = []
        for my in range((h + 15) // 16):
            for mx in range((w + 15) // 16):
                for y in range(my * 16, min(h, my * 16 + 16)):
                    if r0[y * w + mx * 16:y * w + min(w, mx * 16 + 16)] != r1[y * w + mx * 16:y * w + min(w, mx * 16 + 16)]:
                        bad.append((mx, my)); break
        print("luma MBs with different recon (x,y):", bad[:40], "... total", len(bad))
        import ctypes as C, struct
        mbw, mbh = (w + 15) // 16, (h + 15) // 16
        recs = []
        for e in encs:
            buf = (C.c_uint8 * (960 * mbw * mbh))()
            e._lib.WelsHipDebugGetMbRecords.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
            assert e._lib.WelsHipDebugGetMbRecords(e._h, buf, len(buf)) == 0
            recs.append(bytes(buf))
        shown = 0
        for xy in range(mbw * mbh):
            a, b = recs[0][xy * 960:(xy + 1) * 960], recs[1][xy * 960:(xy + 1) * 960]
            if a != b:
                def side(r):
                    mvd = struct.unpack("<32h", r[32:96])
                    return dict(type=r[0], cbp=r[1], mvd=mvd[:2] + mvd[4:6] + mvd[16:18] + mvd[20:22], cost=struct.unpack("<i", r[120:124])[0], nzc=list(r[96:120]))
                print("MB (%d,%d): emu %s" % (xy % mbw, xy // mbw, side(a)))
                print("          hip %s" % (side(b),))
                shown += 1
                if shown >= 6: break
        break
else:
    print("identical for %d frames" % n)
