"""Layer-3 leaf primitives on the MI355X vs the oracle (oracle/prims, pinned against the reference's
_c functions by tests/test_oracle_prims.py).  Bit-exact equality, inputs drawn like the reference's
own unit tests draw them (random u8 pixels, full-range int16 for the quantiser)."""
import ctypes as C
import os

import numpy as np
import pytest

import openh264_amd as oh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC = os.path.join(ROOT, "oracle", "liboracle_prims.so")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(ORC), reason="oracle not built")]
BW = [16, 16, 8, 8, 4, 8, 4]
BH = [16, 8, 16, 8, 4, 4, 8]


@pytest.fixture(scope="module")
def L(hip_lib):
    return oh.load_library(hip_lib), C.CDLL(ORC)


def u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def i16(a):
    return a.ctypes.data_as(C.POINTER(C.c_int16))


def i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def at(a, off):
    return C.cast(a.ctypes.data + int(off), C.POINTER(C.c_uint8))


def planes(rng, n):
    p1 = rng.integers(0, 256, (96, 128), dtype=np.uint8)
    p2 = rng.integers(0, 256, (96, 160), dtype=np.uint8)
    o1 = (rng.integers(8, 70, n) * 128 + rng.integers(8, 100, n)).astype(np.int32)
    o2 = (rng.integers(8, 70, n) * 160 + rng.integers(8, 130, n)).astype(np.int32)
    return p1, p2, o1, o2


def test_sad_satd_sad4(L):
    lib, orc = L
    rng = np.random.default_rng(11)
    n = 3000
    p1, p2, o1, o2 = planes(rng, n)
    for blk in range(7):
        for name, fn, k in (("sad", lib.WelsHipPrimSampleSad, 1), ("satd", lib.WelsHipPrimSampleSatd, 1), ("sad_four", lib.WelsHipPrimSample4Sad, 4)):
            out = np.zeros(n * k, np.int32)
            assert fn(blk, n, u8(p1), C.c_size_t(p1.size), 128, i32(o1), u8(p2), C.c_size_t(p2.size), 160, i32(o2), i32(out)) == 0
            for i in range(0, n, 7):
                if k == 1:
                    assert out[i] == getattr(orc, "orc_" + name)(blk, at(p1, o1[i]), 128, at(p2, o2[i]), 160)
                else:
                    r = (C.c_int32 * 4)()
                    orc.orc_sad_four(blk, at(p1, o1[i]), 128, at(p2, o2[i]), 160, r)
                    assert list(out[i * 4:i * 4 + 4]) == list(r)


def test_dct_quant_scan_dequant_idct(L):
    lib, orc = L
    rng = np.random.default_rng(12)
    n = 4000
    p1, p2, o1, o2 = planes(rng, n)
    dct = np.zeros((n, 16), np.int16)
    assert lib.WelsHipPrimDctT4(n, u8(p1), C.c_size_t(p1.size), 128, i32(o1), u8(p2), C.c_size_t(p2.size), 160, i32(o2), i16(dct)) == 0
    ref = np.zeros(16, np.int16)
    for i in range(0, n, 5):
        orc.orc_dct4x4(i16(ref), at(p1, o1[i]), 128, at(p2, o2[i]), 160)
        assert (dct[i] == ref).all()
    for intra in (0, 1):
        x = rng.integers(-32768, 32768, (n, 16)).astype(np.int16)     # EncUT_EncoderMbAux.cpp:440-444 domain
        x[: n // 2] = dct[: n // 2]
        qp = rng.integers(0, 52, n).astype(np.uint8)
        io = x.copy()
        mx, zz, za = np.zeros(n, np.int16), np.zeros((n, 16), np.int16), np.zeros((n, 16), np.int16)
        ctr, nz = np.zeros(n, np.int32), np.zeros(n, np.int32)
        assert lib.WelsHipPrimQuant4x4(n, i16(io), u8(qp), intra, i16(mx), i16(zz), i16(za), i32(ctr), i32(nz)) == 0
        for i in range(0, n, 3):
            r = x[i].copy()
            m = orc.orc_quant4x4_max(i16(r), int(qp[i]), intra)
            assert (io[i] == r).all() and mx[i] == m
            a, b = np.zeros(16, np.int16), np.zeros(16, np.int16)
            orc.orc_scan4x4_dcac(i16(a), i16(r)); orc.orc_scan4x4_ac(i16(b), i16(r))
            assert (zz[i] == a).all() and (za[i] == b).all()
            assert ctr[i] == orc.orc_single_ctr4x4(i16(a)) and nz[i] == orc.orc_nonzero_count(i16(a))
    lev = rng.integers(-2000, 2000, (n, 16)).astype(np.int16)
    lev[::4] = rng.integers(-32768, 32768, (len(lev[::4]), 16)).astype(np.int16)    # int16 wrap-around cases
    qp = rng.integers(0, 52, n).astype(np.uint8)
    pred = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    rec, deq = np.zeros((n, 16), np.uint8), np.zeros((n, 16), np.int16)
    assert lib.WelsHipPrimDequantIDctRec(n, i16(lev), u8(qp), u8(pred), u8(rec), i16(deq)) == 0
    for i in range(0, n, 3):
        d = lev[i].copy()
        orc.orc_dequant4x4(i16(d), int(qp[i]))
        assert (deq[i] == d).all()
        r = np.zeros(16, np.uint8)
        orc.orc_idct4x4_rec(u8(r), 4, u8(pred[i]), 4, i16(d))
        assert (rec[i] == r).all()


def test_intra_predictors(L):
    lib, orc = L
    rng = np.random.default_rng(13)
    n = 900
    pl = rng.integers(0, 256, (96, 128), dtype=np.uint8)
    off = (rng.integers(8, 70, n) * 128 + rng.integers(8, 90, n)).astype(np.int32)
    mode = rng.integers(0, 9, n).astype(np.uint8)
    avail = rng.integers(0, 4, n).astype(np.uint8)
    pred = np.zeros((n, 16), np.uint8)
    assert lib.WelsHipPrimIntraPred4x4(n, u8(pl), C.c_size_t(pl.size), 128, i32(off), u8(mode), u8(avail), u8(pred)) == 0
    r = np.zeros(16, np.uint8)
    for i in range(n):
        m = int(mode[i])
        if m == 2:
            m = {3: 2, 1: 9, 2: 10, 0: 11}[int(avail[i])]      # I4_PRED_DC / DC_L / DC_T / DC_128
        orc.orc_pred_i4x4(m, u8(r), at(pl, off[i]), 128)
        assert (pred[i] == r).all(), (i, mode[i], avail[i])
    n = 300
    offy = (rng.integers(8, 60, n) * 128 + rng.integers(8, 90, n)).astype(np.int32)
    offc = (rng.integers(8, 70, n) * 128 + rng.integers(8, 90, n)).astype(np.int32)
    m16 = rng.integers(0, 7, n).astype(np.uint8)
    mc = rng.integers(0, 7, n).astype(np.uint8)
    p16, pc = np.zeros((n, 256), np.uint8), np.zeros((n, 128), np.uint8)
    assert lib.WelsHipPrimIntraPredMb(n, u8(pl), C.c_size_t(pl.size), 128, i32(offy), u8(pl), C.c_size_t(pl.size), 128, i32(offc), u8(m16), u8(mc), u8(p16), u8(pc)) == 0
    r16, rc = np.zeros(256, np.uint8), np.zeros(64, np.uint8)
    for i in range(n):
        orc.orc_pred_i16x16(int(m16[i]), u8(r16), at(pl, offy[i]), 128)
        assert (p16[i] == r16).all(), (i, m16[i])
        for p in range(2):
            orc.orc_pred_chroma(int(mc[i]), u8(rc), at(pl, offc[i] + 16 * p), 128)
            assert (pc[i, p * 64:(p + 1) * 64] == rc).all(), (i, mc[i], p)


def test_motion_compensation(L):
    lib, orc = L
    rng = np.random.default_rng(14)
    pl = rng.integers(0, 256, (96, 128), dtype=np.uint8)
    for (w, h, chroma) in [(16, 16, 0), (16, 8, 0), (8, 16, 0), (8, 8, 0), (4, 4, 0), (8, 8, 1), (4, 4, 1), (2, 2, 1), (8, 4, 1)]:
        n = 400
        off = (rng.integers(8, 60, n) * 128 + rng.integers(8, 90, n)).astype(np.int32)
        mv = rng.integers(-64, 64, (n, 2)).astype(np.int16)
        dst = np.zeros((n, h, w), np.uint8)
        assert lib.WelsHipPrimMc(n, u8(pl), C.c_size_t(pl.size), 128, i32(off), i16(mv), w, h, chroma, u8(dst)) == 0
        r = np.zeros((h, w), np.uint8)
        for i in range(n):
            (orc.orc_mc_chroma if chroma else orc.orc_mc_luma)(at(pl, off[i]), 128, u8(r), w, int(mv[i, 0]), int(mv[i, 1]), w, h)
            assert (dst[i] == r).all(), (w, h, chroma, i)


def test_deblock_edges(L):
    lib, orc = L
    rng = np.random.default_rng(15)
    tc0 = None
    for chroma in (0, 1):
        for vertical_edge in (0, 1):
            n = 64
            base = rng.integers(40, 200, (n * 24 + 16, 64)).astype(np.int32)
            pl = (base + rng.integers(-6, 7, base.shape)).clip(0, 255).astype(np.uint8)
            off = (np.arange(n) * 24 * 64 + 8 * 64 + 8).astype(np.int32)      # disjoint 24-row bands
            bs = rng.integers(0, 5, (n, 4)).astype(np.uint8)
            bs[bs[:, 0] == 4] = 4                                              # bS 4 applies to a whole edge
            bs[(bs == 4) & (bs[:, :1] != 4)] = 3
            ia = rng.integers(0, 52, n).astype(np.uint8)
            dev = pl.copy()
            assert lib.WelsHipPrimDeblockEdges(n, u8(dev), C.c_size_t(dev.size), 64, i32(off), vertical_edge, chroma, u8(bs), u8(ia)) == 0
            ref = pl.copy()
            import re
            hdr = open(os.path.join(ROOT, "openh264_amd", "csrc", "common", "h264_tables.h")).read()
            def tab(name):
                body = re.search(r"%s\[[^\]]*\] = \{(.*?)\};" % name, hdr, re.S).group(1)
                return [int(v) for v in body.replace("\n", " ").split(",") if v.strip()]
            alpha, beta, tc = tab("kWhAlpha"), tab("kWhBeta"), tab("kWhTc0")
            for e in range(n):
                a, b = alpha[ia[e]], beta[ia[e]]
                if not (a | b):
                    continue
                p = at(ref, off[e])
                if bs[e, 0] == 4:
                    (orc.orc_deblock_chroma_eq4 if chroma else orc.orc_deblock_luma_eq4)(p, 64, vertical_edge, a, b)
                else:
                    t = (C.c_int8 * 4)(*[(tc[ia[e] * 3 + int(v) - 1] + chroma) if v > 0 else (-1 if not chroma else 0) for v in bs[e]])
                    (orc.orc_deblock_chroma_lt4 if chroma else orc.orc_deblock_luma_lt4)(p, 64, vertical_edge, a, b, t)
            assert (dev == ref).all(), (chroma, vertical_edge)


def test_vaa_sad(L):
    lib, orc = L
    rng = np.random.default_rng(16)
    cur = rng.integers(0, 256, (64, 128), dtype=np.uint8)
    ref = rng.integers(0, 256, (64, 128), dtype=np.uint8)
    n = 24
    off = np.array([(i // 8) * 16 * 128 + (i % 8) * 16 for i in range(n)], np.int32)
    out = np.zeros((n, 4), np.int32)
    assert lib.WelsHipPrimVaaSad8x8(n, u8(cur), u8(ref), C.c_size_t(cur.size), 128, i32(off), i32(out)) == 0
    r = (C.c_int32 * 4)()
    for i in range(n):
        orc.orc_vaa_sad8x8(at(cur, off[i]), at(ref, off[i]), 128, r)
        assert list(out[i]) == list(r)
