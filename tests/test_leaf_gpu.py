"""Layer (3c): the leaf primitives with the reference's own function-pointer signatures (include/welship_leaf.h) on the MI355X.

Two checks.  (1) Every export, called with the (pointer, stride) pairs the reference passes, against the oracle's restatement of the
slot's C function (oracle/prims, pinned against the reference's `_c` functions by tests/test_oracle_prims.py); the quantisers get
the rows of the reference's tables (against the oracle's QP-indexed quantisers) and, beyond what any QP produces, arbitrary FF / MF rows
(against the arithmetic restated in numpy here).
(2) The slots really are installable: the unmodified encoder loop of the reference, built with the patch, runs with all of them in
its dispatch table (WELS_HIP_LEAVES=1: integration/welship_hooks.cpp InstallLeaves -- the static_asserts there tie every export to its
slot's typedef) and must write the bitstream the C functions give, for I and P pictures, CAVLC and CABAC, with the deblocking filter on.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import openh264_amd as oh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC = os.path.join(ROOT, "oracle", "liboracle_prims.so")
REF = os.path.join(ROOT, "oracle", "_ref")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(ORC), reason="oracle not built")]
BLK = ["16x16", "16x8", "8x16", "8x8", "4x4", "8x4", "4x8"]
BW = [16, 16, 8, 8, 4, 8, 4]
BH = [16, 8, 16, 8, 4, 4, 8]


@pytest.fixture(scope="module")
def L(hip_lib):
    lib = C.CDLL(hip_lib)
    assert lib.WelsHipLeafAvailable() == 0
    return lib, C.CDLL(ORC)


def at(a, off=0):
    return C.cast(a.ctypes.data + int(off), C.POINTER(C.c_uint8))


def p16(a, off=0):
    return C.cast(a.ctypes.data + int(off) * 2, C.POINTER(C.c_int16))


def test_sad_satd_sad_four(L):
    lib, orc = L
    rng = np.random.default_rng(5)
    p1 = rng.integers(0, 256, (64, 96), dtype=np.uint8)
    p2 = rng.integers(0, 256, (64, 160), dtype=np.uint8)
    for blk in range(7):
        for _ in range(300):
            o1 = int(rng.integers(4, 40)) * 96 + int(rng.integers(4, 70))
            o2 = int(rng.integers(4, 40)) * 160 + int(rng.integers(4, 130))
            for kind in ("Sad", "Satd"):
                f = getattr(lib, "WelsHipSample%s%s" % (kind, BLK[blk]))
                f.restype = C.c_int32
                assert f(at(p1, o1), 96, at(p2, o2), 160) == getattr(orc, "orc_" + kind.lower())(blk, at(p1, o1), 96, at(p2, o2), 160)
            got, want = (C.c_int32 * 4)(), (C.c_int32 * 4)()
            getattr(lib, "WelsHipSampleSadFour" + BLK[blk])(at(p1, o1), 96, at(p2, o2), 160, got)
            orc.orc_sad_four(blk, at(p1, o1), 96, at(p2, o2), 160, want)
            assert list(got) == list(want)


def _quant_np(d, ff, mf):
    a = np.abs(d.astype(np.int32))
    q = (((ff.astype(np.int32) + a) * mf.astype(np.int32)) >> 16).astype(np.int16).astype(np.int32)
    return np.where(d < 0, -q, q).astype(np.int16), q.astype(np.int16)


def test_transform_and_quantisation_slots(L):
    lib, orc = L
    rng = np.random.default_rng(6)
    p1 = rng.integers(0, 256, (32, 64), dtype=np.uint8)
    p2 = rng.integers(0, 256, (32, 48), dtype=np.uint8)
    for _ in range(200):
        o1, o2 = int(rng.integers(0, 20)) * 64 + int(rng.integers(0, 50)), int(rng.integers(0, 20)) * 48 + int(rng.integers(0, 36))
        got, want = np.zeros(64, np.int16), np.zeros(64, np.int16)
        lib.WelsHipDctT4(p16(got), at(p1, o1), 64, at(p2, o2), 48)
        orc.orc_dct4x4(p16(want), at(p1, o1), 64, at(p2, o2), 48)
        assert (got[:16] == want[:16]).all()
        lib.WelsHipDctFourT4(p16(got), at(p1, o1), 64, at(p2, o2), 48)
        for b, (dx, dy) in enumerate(((0, 0), (4, 0), (0, 4), (4, 4))):     # encode_mb_aux.cpp:348-368: left, right, lower left, lower right
            orc.orc_dct4x4(p16(want, b * 16), at(p1, o1 + dy * 64 + dx), 64, at(p2, o2 + dy * 48 + dx), 48)
        assert (got == want).all()
    for _ in range(300):
        d = rng.integers(-32768, 32768, 64).astype(np.int16)
        if _ % 3 == 0:
            d = rng.integers(-600, 600, 64).astype(np.int16)
        ff = rng.integers(0, 4000, 8).astype(np.int16)
        mf = rng.integers(1, 14000, 8).astype(np.int16)
        idx = np.arange(64) & 7
        want, wabs = _quant_np(d, ff[idx], mf[idx])
        g = d.copy(); lib.WelsHipQuant4x4(p16(g), p16(ff), p16(mf)); assert (g[:16] == want[:16]).all() and (g[16:] == d[16:]).all()
        g = d.copy(); lib.WelsHipQuantFour4x4(p16(g), p16(ff), p16(mf)); assert (g == want).all()
        g = d.copy(); mx = np.zeros(4, np.int16); lib.WelsHipQuantFour4x4Max(p16(g), p16(ff), p16(mf), p16(mx))
        assert (g == want).all() and list(mx) == [int(max(0, wabs[b * 16:b * 16 + 16].max())) for b in range(4)]
        # ... and with the rows the encoder really passes (the tables of encode_mb_aux.cpp:39-157), against the oracle's QP-indexed quantisers
        qp, intra = int(rng.integers(0, 52)), int(rng.integers(0, 2))
        tff, tmf = np.zeros(8, np.int16), np.zeros(8, np.int16)
        orc.orc_quant_rows(qp, intra, p16(tff), p16(tmf))
        w = d.copy()
        wmx = [orc.orc_quant4x4_max(p16(w, b * 16), qp, intra) for b in range(4)]
        g = d.copy(); lib.WelsHipQuant4x4(p16(g), p16(tff), p16(tmf)); assert (g[:16] == w[:16]).all()
        g = d.copy(); lib.WelsHipQuantFour4x4(p16(g), p16(tff), p16(tmf)); assert (g == w).all()
        g = d.copy(); mx = np.zeros(4, np.int16); lib.WelsHipQuantFour4x4Max(p16(g), p16(tff), p16(tmf), p16(mx)); assert (g == w).all() and [int(x) for x in mx] == wmx
        g = d.copy(); lib.WelsHipQuant4x4Dc(p16(g), C.c_int16(int(ff[0])), C.c_int16(int(mf[0])))
        w = d.copy(); orc.orc_quant4x4_dc(p16(w), C.c_int16(int(ff[0])), C.c_int16(int(mf[0]))); assert (g == w).all()
        # chroma DC: Hadamard + quantisation / the skip test
        rs = rng.integers(-2000, 2000, 64).astype(np.int16); rs2 = rs.copy()
        dg, bg, dw, bw = (np.zeros(4, np.int16) for _ in range(4))
        lib.WelsHipHadamardQuant2x2.restype = C.c_int32
        r1 = lib.WelsHipHadamardQuant2x2(p16(rs), C.c_int16(int(ff[0])), C.c_int16(int(mf[0])), p16(dg), p16(bg))
        r2 = orc.orc_hadamard_quant2x2(p16(rs2), C.c_int16(int(ff[0])), C.c_int16(int(mf[0])), p16(dw), p16(bw))
        assert r1 == r2 and (rs == rs2).all() and (dg == dw).all() and (bg == bw).all()
        rs = rng.integers(-60, 60, 64).astype(np.int16)
        assert lib.WelsHipHadamardQuant2x2Skip(p16(rs), C.c_int16(int(ff[0])), C.c_int16(int(max(mf[0], 2000)))) == \
            orc.orc_hadamard_quant2x2_skip(p16(rs), C.c_int16(int(ff[0])), C.c_int16(int(max(mf[0], 2000))))
        # luma DC Hadamard, scans, scores
        big = rng.integers(-3000, 3000, 256).astype(np.int16)
        g, w = np.zeros(16, np.int16), np.zeros(16, np.int16)
        lib.WelsHipHadamardT4Dc(p16(g), p16(big)); orc.orc_hadamard_t4_dc(p16(w), p16(big)); assert (g == w).all()
        lv = (rng.integers(-3, 4, 16) * (rng.random(16) < 0.4)).astype(np.int16)
        lib.WelsHipScan4x4DcAc(p16(g), p16(lv)); orc.orc_scan4x4_dcac(p16(w), p16(lv)); assert (g == w).all()
        lib.WelsHipScan4x4Ac(p16(g), p16(lv)); orc.orc_scan4x4_ac(p16(w), p16(lv)); assert (g == w).all()
        assert lib.WelsHipCalculateSingleCtr4x4(p16(lv)) == orc.orc_single_ctr4x4(p16(lv))
        assert lib.WelsHipGetNoneZeroCount(p16(lv)) == orc.orc_nonzero_count(p16(lv))


def test_reconstruction_slots(L):
    lib, orc = L
    rng = np.random.default_rng(7)
    for _ in range(200):
        res = rng.integers(-2000, 2000, 64).astype(np.int16)
        mf = rng.integers(10, 300, 8).astype(np.uint16)
        g = res.copy(); lib.WelsHipDequant4x4(p16(g), mf.ctypes.data_as(C.POINTER(C.c_uint16)))
        want = (res.astype(np.int32) * mf[np.arange(64) & 7].astype(np.int32)).astype(np.int16)
        assert (g[:16] == want[:16]).all() and (g[16:] == res[16:]).all()
        g = res.copy(); lib.WelsHipDequantFour4x4(p16(g), mf.ctypes.data_as(C.POINTER(C.c_uint16))); assert (g == want).all()
        # luma DC: inverse Hadamard x the factor (the qp >= 12 branch the table slot holds, decode_mb_aux.cpp:120-152): restated
        g = res[:16].copy(); lib.WelsHipDequantIHadamard4x4(p16(g), C.c_uint16(int(mf[0])))
        t = res[:16].astype(np.int32).reshape(4, 4)
        def had(v):
            a, b, c, d = v[..., 0] + v[..., 2], v[..., 0] - v[..., 2], v[..., 1] - v[..., 3], v[..., 1] + v[..., 3]
            return np.stack([a + d, b + c, b - c, a - d], -1).astype(np.int16).astype(np.int32)
        h = had(had(t).T).T
        assert (g == (h * int(mf[0])).astype(np.int16).reshape(16)).all()
        rec = np.zeros((24, 40), np.uint8); pred = rng.integers(0, 256, (24, 32), dtype=np.uint8)
        co = rng.integers(-900, 900, 64).astype(np.int16)
        want = rec.copy()
        lib.WelsHipIDctT4Rec(at(rec, 3 * 40 + 5), 40, at(pred, 2 * 32 + 7), 32, p16(co))
        orc.orc_idct4x4_rec(at(want, 3 * 40 + 5), 40, at(pred, 2 * 32 + 7), 32, p16(co))
        assert (rec == want).all()
        lib.WelsHipIDctFourT4Rec(at(rec, 9 * 40 + 12), 40, at(pred, 8 * 32 + 3), 32, p16(co))
        for b, (dx, dy) in enumerate(((0, 0), (4, 0), (0, 4), (4, 4))):
            orc.orc_idct4x4_rec(at(want, (9 + dy) * 40 + 12 + dx), 40, at(pred, (8 + dy) * 32 + 3 + dx), 32, p16(co, b * 16))
        assert (rec == want).all()
        rec = np.zeros((24, 40), np.uint8)
        lib.WelsHipIDctRecI16x16Dc(at(rec, 4 * 40 + 8), 40, at(pred, 3 * 32 + 2), 32, p16(co))
        yy, xx = np.mgrid[0:16, 0:16]
        w = np.clip(pred[3:19, 2:18].astype(np.int32) + ((co[(yy & 12) + (xx >> 2)].astype(np.int32) + 32) >> 6), 0, 255)
        assert (rec[4:20, 8:24] == w).all() and rec[:4].sum() == 0 and rec[20:].sum() == 0 and rec[:, :8].sum() == 0 and rec[:, 24:].sum() == 0


def test_motion_compensation_slots(L):
    lib, orc = L
    rng = np.random.default_rng(8)
    src = rng.integers(0, 256, (64, 96), dtype=np.uint8)
    for chroma, sizes in ((0, ((16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4))), (1, ((8, 8), (8, 4), (4, 8), (4, 4), (2, 4), (4, 2), (2, 2)))):
        for (w, h) in sizes:
            for _ in range(100):
                o = int(rng.integers(8, 36)) * 96 + int(rng.integers(8, 70))
                mvx, mvy = int(rng.integers(-40, 40)), int(rng.integers(-40, 40))
                got = np.full((20, 24), 7, np.uint8); want = got.copy()
                f = lib.WelsHipMcChroma if chroma else lib.WelsHipMcLuma
                f(at(src, o), 96, at(got, 2 * 24 + 3), 24, C.c_int16(mvx), C.c_int16(mvy), w, h)
                (orc.orc_mc_chroma if chroma else orc.orc_mc_luma)(at(src, o), 96, at(want, 2 * 24 + 3), 24, mvx, mvy, w, h)
                assert (got == want).all(), (chroma, w, h, mvx, mvy)
    # the half-sample planes of the refinement (widths / heights 4..17) = the (2,0) / (0,2) / (2,2) positions of the quarter-sample function
    for name, mv in (("WelsHipMcHorVer20", (2, 0)), ("WelsHipMcHorVer02", (0, 2)), ("WelsHipMcHorVer22", (2, 2))):
        for (w, h) in ((16, 16), (17, 16), (16, 17), (17, 17), (9, 8), (8, 9), (5, 4), (4, 5), (9, 17)):
            o = int(rng.integers(8, 36)) * 96 + int(rng.integers(8, 70))
            got = np.full((20, 24), 9, np.uint8); want = got.copy()
            getattr(lib, name)(at(src, o), 96, at(got, 24 + 2), 24, w, h)
            for y in range(h):       # the oracle's function takes the block sizes of the encoder: sample by sample here
                for x in range(w):
                    one = np.zeros(16, np.uint8)
                    orc.orc_mc_luma(at(src, o + y * 96 + x - (x % 4) - (y % 4) * 96), 96, at(one), 4, mv[0], mv[1], 4, 4)
                    want[1 + y, 2 + x] = one[(y % 4) * 4 + (x % 4)]
            assert (got == want).all(), (name, w, h)
    a = rng.integers(0, 256, (20, 32), dtype=np.uint8); b = rng.integers(0, 256, (20, 48), dtype=np.uint8)
    got = np.zeros((20, 24), np.uint8)
    lib.WelsHipPixelAvg(at(got, 24 + 1), 24, at(a, 32 + 3), 32, at(b, 2 * 48 + 5), 48, 16, 8)
    assert (got[1:9, 1:17] == ((a[1:9, 3:19].astype(np.int32) + b[2:10, 5:21] + 1) >> 1)).all() and got[0].sum() == 0 and got[9:].sum() == 0


def test_intra_predictor_slots(L):
    lib, orc = L
    rng = np.random.default_rng(9)
    plane = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    # (export, the reference's table index): wels_common_defs.h:330-371
    i4 = (("V", 0), ("H", 1), ("Dc", 2), ("DDL", 3), ("DDR", 4), ("VR", 5), ("HD", 6), ("VL", 7), ("HU", 8), ("DcLeft", 9), ("DcTop", 10), ("DcNA", 11),
          ("DDLTop", 12), ("VLTop", 13))
    i16 = (("V", 0), ("H", 1), ("Dc", 2), ("Plane", 3), ("DcLeft", 4), ("DcTop", 5), ("DcNA", 6))
    ic = (("Dc", 0), ("H", 1), ("V", 2), ("Plane", 3), ("DcLeft", 4), ("DcTop", 5), ("DcNA", 6))
    for fam, tab, n, of in (("WelsHipI4x4LumaPred", i4, 16, orc.orc_pred_i4x4), ("WelsHipI16x16LumaPred", i16, 256, orc.orc_pred_i16x16),
                            ("WelsHipIChromaPred", ic, 64, orc.orc_pred_chroma)):
        for name, mode in tab:
            for _ in range(60):
                o = int(rng.integers(4, 24)) * 64 + int(rng.integers(4, 40))
                got, want = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
                getattr(lib, fam + name)(at(got), at(plane, o), 64)
                of(mode, at(want), at(plane, o), 64)
                assert (got == want).all(), (fam, name)


def test_deblocking_slots(L):
    lib, orc = L
    rng = np.random.default_rng(10)
    for _ in range(400):
        base = rng.integers(60, 200)
        pl = np.clip(base + rng.integers(-14, 15, (40, 48)), 0, 255).astype(np.uint8)
        pl2 = np.clip(base + rng.integers(-14, 15, (40, 48)), 0, 255).astype(np.uint8)
        alpha, beta = int(rng.integers(4, 120)), int(rng.integers(2, 18))
        tc = np.array([int(rng.integers(-1, 9)) for _ in range(4)], np.int8)
        tcp = tc.ctypes.data_as(C.POINTER(C.c_int8))
        o = 12 * 48 + 16
        for hor in (0, 1):        # 0: "V" (horizontal edge), 1: "H" (vertical edge)
            s = "H" if hor else "V"
            g, w = pl.copy(), pl.copy()
            getattr(lib, "WelsHipDeblockLumaLt4" + s)(at(g, o), 48, alpha, beta, tcp); orc.orc_deblock_luma_lt4(at(w, o), 48, hor, alpha, beta, tcp); assert (g == w).all()
            g, w = pl.copy(), pl.copy()
            getattr(lib, "WelsHipDeblockLumaEq4" + s)(at(g, o), 48, alpha, beta); orc.orc_deblock_luma_eq4(at(w, o), 48, hor, alpha, beta); assert (g == w).all()
            g, w, g2, w2 = pl.copy(), pl.copy(), pl2.copy(), pl2.copy()
            getattr(lib, "WelsHipDeblockChromaLt4" + s)(at(g, o), at(g2, o), 48, alpha, beta, tcp)
            orc.orc_deblock_chroma_lt4(at(w, o), 48, hor, alpha, beta, tcp); orc.orc_deblock_chroma_lt4(at(w2, o), 48, hor, alpha, beta, tcp)
            assert (g == w).all() and (g2 == w2).all()
            g, w, g2, w2 = pl.copy(), pl.copy(), pl2.copy(), pl2.copy()
            getattr(lib, "WelsHipDeblockChromaEq4" + s)(at(g, o), at(g2, o), 48, alpha, beta)
            orc.orc_deblock_chroma_eq4(at(w, o), 48, hor, alpha, beta); orc.orc_deblock_chroma_eq4(at(w2, o), 48, hor, alpha, beta)
            assert (g == w).all() and (g2 == w2).all()


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "libref_prims.so")), reason="oracle/_ref not built")
def test_copy_clear_and_combined_intra_cost_slots(L):
    """pfCopy*, pfSetMemZero* and the five pfIntra*Combined3* slots (round 6) against the reference's own `_c` functions (oracle/_ref/libref_prims.so:
    copy_mb.cpp:38-111, sample.cpp:153-331).  The Combined3 functions also leave bytes in the caller's buffers -- compared too."""
    lib, _ = L
    ref = C.CDLL(os.path.join(REF, "libref_prims.so"))
    rng = np.random.default_rng(12)
    for name, w, h in (("4x4", 4, 4), ("8x4", 8, 4), ("4x8", 4, 8), ("8x8", 8, 8), ("16x8", 16, 8), ("8x16", 8, 16), ("16x16", 16, 16)):
        for _ in range(40):
            src = rng.integers(0, 256, (40, 64), dtype=np.uint8)
            got = rng.integers(0, 256, (40, 48), dtype=np.uint8)
            want = got.copy()
            so, do = int(rng.integers(0, 20)) * 64 + int(rng.integers(0, 40)), int(rng.integers(0, 20)) * 48 + int(rng.integers(0, 30))
            getattr(lib, "WelsHipCopy" + name)(at(got, do), 48, at(src, so), 64)
            ref.ref_copy(w, h, at(want, do), 48, at(src, so), 64)
            assert (got == want).all(), name
    for size in (8, 64, 128, 768, 8192, 8192 + 64, 40000):
        got = rng.integers(1, 256, size + 32, dtype=np.uint8)
        want = got.copy()
        lib.WelsHipSetMemZero(C.cast(got.ctypes.data + 16, C.c_void_p), size)
        ref.ref_set_mem_zero(C.cast(want.ctypes.data + 16, C.c_void_p), size)
        assert (got == want).all() and not got[16:16 + size].any() and got[:16].all() and got[16 + size:].all(), size
    lib.WelsHipIntra4x4Combined3Satd.restype = lib.WelsHipIntra16x16Combined3Satd.restype = lib.WelsHipIntra16x16Combined3Sad.restype = C.c_int32
    lib.WelsHipIntra8x8Combined3Satd.restype = lib.WelsHipIntra8x8Combined3Sad.restype = C.c_int32
    for case in range(120):
        flat = case % 3 == 0       # near-flat blocks: equal costs, where the order of the candidates decides
        dec = (rng.integers(100, 104, (40, 64)) if flat else rng.integers(0, 256, (40, 64))).astype(np.uint8)
        dec2 = (rng.integers(100, 104, (40, 64)) if flat else rng.integers(0, 256, (40, 64))).astype(np.uint8)
        enc = (rng.integers(100, 104, (40, 48)) if flat else rng.integers(0, 256, (40, 48))).astype(np.uint8)
        enc2 = (rng.integers(100, 104, (40, 48)) if flat else rng.integers(0, 256, (40, 48))).astype(np.uint8)
        o, e = 8 * 64 + 16, 4 * 48 + 8
        lam = int(rng.integers(0, 40))
        gm, wm = C.c_int32(-7), C.c_int32(-7)
        gd, wd = np.full(16, 9, np.uint8), np.full(16, 9, np.uint8)
        l2, l1, l0 = (int(x) for x in rng.integers(0, 60, 3))
        g = lib.WelsHipIntra4x4Combined3Satd(at(dec, o), 64, at(enc, e), 48, at(gd), C.byref(gm), l2, l1, l0)
        w_ = ref.ref_intra4x4_combined3_satd(at(dec, o), 64, at(enc, e), 48, at(wd), C.byref(wm), l2, l1, l0)
        assert (g, gm.value) == (w_, wm.value) and (gd == wd).all(), ("4x4", case)
        for satd in (1, 0):
            gd, wd = np.full(256, 9, np.uint8), np.full(256, 9, np.uint8)
            f = lib.WelsHipIntra16x16Combined3Satd if satd else lib.WelsHipIntra16x16Combined3Sad
            g = f(at(dec, o), 64, at(enc, e), 48, C.byref(gm), lam, at(gd))
            w_ = ref.ref_intra16x16_combined3(satd, at(dec, o), 64, at(enc, e), 48, C.byref(wm), lam, at(wd))
            assert (g, gm.value) == (w_, wm.value) and (gd == wd).all(), ("16x16", satd, case)
            gd, wd = np.full(128, 9, np.uint8), np.full(128, 9, np.uint8)
            f = lib.WelsHipIntra8x8Combined3Satd if satd else lib.WelsHipIntra8x8Combined3Sad
            g = f(at(dec, o), 64, at(enc, e), 48, C.byref(gm), lam, at(gd), at(dec2, o), at(enc2, e))
            w_ = ref.ref_intra8x8_combined3(satd, at(dec, o), 64, at(enc, e), 48, C.byref(wm), lam, at(wd), at(dec2, o), at(enc2, e))
            assert (g, gm.value) == (w_, wm.value) and (gd == wd).all(), ("8x8", satd, case)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ref_enc_hip")), reason="oracle/_ref (hooked reference) not built")
@pytest.mark.parametrize("leaves", ["1", "2"])
@pytest.mark.parametrize("extra", [["-rc", "-1", "-qp", "28"], ["-rc", "-1", "-qp", "20", "-cabac", "1", "-profile", "77", "-slcmd", "1", "-slcnum", "2"],
                                   ["-rc", "1", "-bitrate", "90000", "-complexity", "2"]])
def test_reference_encoder_loop_on_the_leaf_slots(hip_lib, tmp_path, extra, leaves):
    """The reference's own macroblock loops with every leaf slot of its dispatch table served by the device: byte-identical stream.
    leaves = 2 also fills the five Combined3 slots, which the C build leaves NULL: mode decision then takes its combined paths (same decisions)."""
    from openh264_amd.utils.synth import synth_sequence
    w, h, n = 80, 64, 4
    src = str(tmp_path / "c.yuv")
    open(src, "wb").write(synth_sequence(w, h, n))
    outs = []
    for exe, env in (("ref_enc", dict(os.environ)), ("ref_enc_hip", dict(os.environ, WELSHIP_LIB=hip_lib, WELS_HIP_LEAVES=leaves, WELS_HIP_TRACE="1"))):
        out = str(tmp_path / (exe + ".264"))
        p = subprocess.run([os.path.join(REF, exe), "-i", src, "-w", str(w), "-h", str(h), "-o", out, "-quiet", "-threads", "1", "-deblock", "0"] + extra, env=env,
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
        err = p.stderr.decode(errors="replace")
        assert p.returncode == 0, err[-2000:]
        if exe == "ref_enc_hip":
            assert ("%d leaf functions installed" % (93 if leaves == "1" else 98)) in err, err[-2000:]
        outs.append(open(out, "rb").read())
    assert len(outs[0]) > 200 and outs[0] == outs[1]
