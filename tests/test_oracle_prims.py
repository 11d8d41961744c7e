"""Pins the plain-C restatement of the primitives (oracle/prims, liboracle_prims.so) against the
reference's own C fallback (oracle/_ref/libref_prims.so = trampolines onto the real `_c` functions).
Input domains follow the reference's unit tests (test/encoder/EncUT_*.cpp): random u8 pixels, quant
inputs over the full int16 range.  Runs in the build container (the reference does not travel); on a
box without oracle/_ref it is skipped -- the golden bitstream tests still pin the frame level there."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC = os.path.join(ROOT, "oracle", "liboracle_prims.so")
REF = os.path.join(ROOT, "oracle", "_ref", "libref_prims.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(ORC) and os.path.exists(REF)), reason="oracle libs not built")

BW = [16, 16, 8, 8, 4, 8, 4]
BH = [16, 8, 16, 8, 4, 4, 8]


@pytest.fixture(scope="module")
def libs():
    o, r = C.CDLL(ORC), C.CDLL(REF)
    return o, r


def u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def i16p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int16))


def both(libs, name):
    o, r = libs
    return getattr(o, "orc_" + name), getattr(r, "ref_" + name)


def test_sad_satd(libs):
    rng = np.random.default_rng(1)
    for blk in range(7):
        for _ in range(200):
            a = rng.integers(0, 256, (40, 48), dtype=np.uint8)
            b = rng.integers(0, 256, (40, 64), dtype=np.uint8)
            pa, pb = C.cast(a.ctypes.data + 48 * 4 + 4, C.POINTER(C.c_uint8)), C.cast(b.ctypes.data + 64 * 4 + 4, C.POINTER(C.c_uint8))
            for name in ("sad", "satd"):
                fo, fr = both(libs, name)
                assert fo(blk, pa, 48, pb, 64) == fr(blk, pa, 48, pb, 64)
            fo, fr = both(libs, "sad_four")
            oo, rr = (C.c_int32 * 4)(), (C.c_int32 * 4)()
            fo(blk, pa, 48, pb, 64, oo)
            fr(blk, pa, 48, pb, 64, rr)
            assert list(oo) == list(rr)


def test_transform_quant(libs):
    rng = np.random.default_rng(2)
    for it in range(400):
        p1 = rng.integers(0, 256, (4, 20), dtype=np.uint8)
        p2 = rng.integers(0, 256, (4, 16), dtype=np.uint8)
        do, dr = np.zeros(16, np.int16), np.zeros(16, np.int16)
        fo, fr = both(libs, "dct4x4")
        fo(i16p(do), u8p(p1), 20, u8p(p2), 16)
        fr(i16p(dr), u8p(p1), 20, u8p(p2), 16)
        assert (do == dr).all()
        qp, intra = int(rng.integers(0, 52)), int(rng.integers(0, 2))
        # full int16 domain like EncUT_EncoderMbAux.cpp:440-444
        x = rng.integers(-32768, 32768, 16).astype(np.int16)
        for name in ("quant4x4",):
            a, b = x.copy(), x.copy()
            fo, fr = both(libs, name)
            fo(i16p(a), qp, intra); fr(i16p(b), qp, intra)
            assert (a == b).all()
        a, b = x.copy(), x.copy()
        fo, fr = both(libs, "quant4x4_max")
        assert fo(i16p(a), qp, intra) == fr(i16p(b), qp, intra) and (a == b).all()
        ff, mf = int(rng.integers(0, 1600)), int(rng.integers(1, 32768))
        a, b = x.copy(), x.copy()
        fo, fr = both(libs, "quant4x4_dc")
        fo(i16p(a), ff, mf); fr(i16p(b), ff, mf)
        assert (a == b).all()
        # levels: scans / score / count
        lv = rng.integers(-3, 4, 16).astype(np.int16) * (rng.integers(0, 4, 16) == 0)
        lv = lv.astype(np.int16)
        for name in ("scan4x4_dcac", "scan4x4_ac"):
            a, b = np.zeros(16, np.int16), np.zeros(16, np.int16)
            fo, fr = both(libs, name)
            fo(i16p(a), i16p(lv)); fr(i16p(b), i16p(lv))
            assert (a == b).all()
        for name in ("single_ctr4x4", "nonzero_count"):
            fo, fr = both(libs, name)
            assert fo(i16p(lv)) == fr(i16p(lv))
        # dequant + idct (coefficients in the range real quantised data produces, plus wrap-around cases)
        co = rng.integers(-2048, 2048, 16).astype(np.int16) if it % 4 else rng.integers(-32768, 32768, 16).astype(np.int16)
        a, b = co.copy(), co.copy()
        fo, fr = both(libs, "dequant4x4")
        fo(i16p(a), qp); fr(i16p(b), qp)
        assert (a == b).all()
        a, b = co.copy(), co.copy()
        fo, fr = both(libs, "dequant_ihadamard4x4")
        fo(i16p(a), qp); fr(i16p(b), qp)
        assert (a == b).all()
        a, b = co[:4].copy(), co[:4].copy()
        fo, fr = both(libs, "dequant_ihadamard2x2_dc")
        fo(i16p(a), qp); fr(i16p(b), qp)
        assert (a == b).all()
        pred = rng.integers(0, 256, (4, 16), dtype=np.uint8)
        ra, rb = np.zeros((4, 24), np.uint8), np.zeros((4, 24), np.uint8)
        fo, fr = both(libs, "idct4x4_rec")
        fo(u8p(ra), 24, u8p(pred), 16, i16p(co)); fr(u8p(rb), 24, u8p(pred), 16, i16p(co))
        assert (ra == rb).all()
        # luma DC hadamard over a 256-coefficient MB + chroma 2x2
        mb = rng.integers(-4080, 4081, 256).astype(np.int16)
        a, b = np.zeros(16, np.int16), np.zeros(16, np.int16)
        fo, fr = both(libs, "hadamard_t4_dc")
        fo(i16p(a), i16p(mb)); fr(i16p(b), i16p(mb))
        assert (a == b).all()
        rs1, rs2 = mb[:64].copy(), mb[:64].copy()
        d1, d2, b1, b2 = (np.zeros(4, np.int16) for _ in range(4))
        fo, fr = both(libs, "hadamard_quant2x2")
        assert fo(i16p(rs1), ff, max(mf >> 1, 1), i16p(d1), i16p(b1)) == fr(i16p(rs2), ff, max(mf >> 1, 1), i16p(d2), i16p(b2))
        assert (rs1 == rs2).all() and (d1 == d2).all() and (b1 == b2).all()
        fo, fr = both(libs, "hadamard_quant2x2_skip")
        assert bool(fo(i16p(mb), ff, max(mf >> 1, 1))) == bool(fr(i16p(mb), ff, max(mf >> 1, 1)))


def test_intra_predictors(libs):
    rng = np.random.default_rng(3)
    for _ in range(100):
        pic = rng.integers(0, 256, (40, 64), dtype=np.uint8)
        ref = C.cast(pic.ctypes.data + 64 * 8 + 16, C.POINTER(C.c_uint8))
        for name, n, size in (("pred_i4x4", 14, 16), ("pred_i16x16", 7, 256), ("pred_chroma", 7, 64)):
            fo, fr = both(libs, name)
            for mode in range(n):
                a, b = np.zeros(size, np.uint8), np.zeros(size, np.uint8)
                fo(mode, u8p(a), ref, 64); fr(mode, u8p(b), ref, 64)
                assert (a == b).all(), (name, mode)


def test_motion_compensation(libs):
    rng = np.random.default_rng(4)
    sizes = [(16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4)]
    for _ in range(40):
        pic = rng.integers(0, 256, (48, 64), dtype=np.uint8)
        src = C.cast(pic.ctypes.data + 64 * 12 + 16, C.POINTER(C.c_uint8))
        for (w, h) in sizes:
            for mvx in range(4):
                for mvy in range(4):
                    a, b = np.zeros((16, 16), np.uint8), np.zeros((16, 16), np.uint8)
                    fo, fr = both(libs, "mc_luma")
                    fo(src, 64, u8p(a), 16, mvx, mvy, w, h); fr(src, 64, u8p(b), 16, mvx, mvy, w, h)
                    assert (a == b).all(), (w, h, mvx, mvy)
        for (w, h) in [(8, 8), (8, 4), (4, 8), (4, 4), (2, 2)]:
            mvx, mvy = int(rng.integers(0, 8)), int(rng.integers(0, 8))
            a, b = np.zeros((8, 8), np.uint8), np.zeros((8, 8), np.uint8)
            fo, fr = both(libs, "mc_chroma")
            fo(src, 64, u8p(a), 8, mvx, mvy, w, h); fr(src, 64, u8p(b), 8, mvx, mvy, w, h)
            assert (a == b).all()


def test_deblock_edges(libs):
    rng = np.random.default_rng(5)
    for it in range(300):
        base = rng.integers(0, 256, dtype=np.uint8)
        pic = (base + rng.integers(-12, 13, (32, 32))).clip(0, 255).astype(np.uint8) if it % 3 else rng.integers(0, 256, (32, 32), dtype=np.uint8)
        alpha, beta = int(rng.integers(0, 256)), int(rng.integers(0, 19))
        tc = (C.c_int8 * 4)(*[int(v) for v in rng.integers(-1, 14, 4)])
        for horizontal in (0, 1):
            for name, has_tc in (("deblock_luma_lt4", True), ("deblock_luma_eq4", False), ("deblock_chroma_lt4", True), ("deblock_chroma_eq4", False)):
                a, b = pic.copy(), pic.copy()
                pa = C.cast(a.ctypes.data + 32 * 8 + 8, C.POINTER(C.c_uint8))
                pb = C.cast(b.ctypes.data + 32 * 8 + 8, C.POINTER(C.c_uint8))
                fo, fr = both(libs, name)
                if has_tc:
                    fo(pa, 32, horizontal, alpha, beta, tc); fr(pb, 32, horizontal, alpha, beta, tc)
                else:
                    fo(pa, 32, horizontal, alpha, beta); fr(pb, 32, horizontal, alpha, beta)
                assert (a == b).all(), (name, horizontal)


def test_vaa_sad(libs):
    rng = np.random.default_rng(6)
    fo, fr = both(libs, "vaa_sad8x8")
    for _ in range(50):
        a = rng.integers(0, 256, (16, 32), dtype=np.uint8)
        b = rng.integers(0, 256, (16, 32), dtype=np.uint8)
        so, sr = (C.c_int32 * 4)(), (C.c_int32 * 4)()
        fo(u8p(a), u8p(b), 32, so); fr(u8p(a), u8p(b), 32, sr)
        assert list(so) == list(sr)


def test_quantiser_table_rows(libs):
    """orc_quant_rows: the rows the encoder passes to pfQuantization* (g_kiQuantInterFF[qp (+ 6)] / g_kiQuantMF[qp], encode_mb_aux.cpp:39-157) --
    what tests/test_leaf_gpu.py feeds the device's quantiser slots."""
    o, r = libs
    for qp in range(52):
        for intra in (0, 1):
            of, om, rf, rm = (np.zeros(8, np.int16) for _ in range(4))
            o.orc_quant_rows(qp, intra, i16p(of), i16p(om))
            r.ref_quant_rows(qp, intra, i16p(rf), i16p(rm))
            assert (of == rf).all() and (om == rm).all(), (qp, intra)
