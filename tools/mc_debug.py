import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openh264_amd as oh
from openh264_amd import build as B
lib = oh.load_library(B.build_hip()); orc = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle_prims.so"))
u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
rng = np.random.default_rng(14)
pl = rng.integers(0, 256, (96, 128), dtype=np.uint8)
n = 64
off = (rng.integers(8, 60, n) * 128 + rng.integers(8, 90, n)).astype(np.int32)
for fy in range(4):
    for fx in range(4):
        mv = np.tile(np.array([fx, fy], np.int16), (n, 1)).copy()
        dst = np.zeros((n, 16, 16), np.uint8)
        assert lib.WelsHipPrimMc(n, u8(pl), C.c_size_t(pl.size), 128, off.ctypes.data_as(C.POINTER(C.c_int32)), mv.ctypes.data_as(C.POINTER(C.c_int16)), 16, 16, 0, u8(dst)) == 0
        bad = 0; ex = None
        r = np.zeros((16, 16), np.uint8)
        for i in range(n):
            orc.orc_mc_luma(C.cast(pl.ctypes.data + int(off[i]), C.POINTER(C.c_uint8)), 128, u8(r), 16, fx, fy, 16, 16)
            d = (dst[i] != r)
            if d.any():
                bad += 1
                if ex is None: ex = (i, int(off[i]) & 3, np.argwhere(d)[:6].tolist(), dst[i][d][:6].tolist(), r[d][:6].tolist())
        print("fx %d fy %d: bad %d/%d %s" % (fx, fy, bad, n, ex))
