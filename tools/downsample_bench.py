"""Throughput of the down-sampling kernels on resident planes (WelsHipDownsampleBench): GB/s of algorithmic bytes against the HBM peak."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openh264_amd as oh
lib = oh.load_library()
lib.WelsHipDownsampleBench.argtypes = [C.c_int] * 8 + [C.POINTER(C.c_double)]
names = {0: "dyadic 2:1", 1: "quarter 4:1", 2: "one third 3:1", 3: "general fast (luma)", 4: "general accurate (chroma)"}
rows = []
for mode, sw, sh, dw, dh, n in [(0, 1920, 1080, 960, 540, 256), (0, 1280, 720, 640, 360, 512), (0, 3840, 2160, 1920, 1080, 64), (1, 1280, 720, 320, 180, 512),
                                (2, 1920, 1080, 640, 360, 256), (3, 1920, 1080, 1280, 720, 256), (4, 960, 540, 640, 360, 512)]:
    out = (C.c_double * 2)()
    rc = lib.WelsHipDownsampleBench(0, mode, n, sw, sh, dw, dh, 20, out)
    gbs = out[1] / (out[0] * 1e-3) / 1e9 if rc == 0 and out[0] > 0 else 0.0
    rows.append({"kernel": names[mode], "src": "%dx%d" % (sw, sh), "dst": "%dx%d" % (dw, dh), "planes": n, "ms_per_launch": out[0], "algorithmic_GB": out[1] / 1e9,
                 "achieved_GBs": gbs, "frac_of_8TBs": gbs / 8000.0, "rc": rc})
    print(json.dumps(rows[-1]))
