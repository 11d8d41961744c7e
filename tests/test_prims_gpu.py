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


N_PAIRS = 131072          # pairs per block size and function: every one compared (SURVEY 7 minimum slice: 10^6 random pairs over the family)


def _sad_family(lib, orc, p1, s1, o1, p2, s2, o2, blocks=range(7)):
    """Every block size x {SAD, SATD, four-neighbour SAD} for the given (enc, ref) pairs: the device's answers against the oracle's, all of them."""
    n = len(o1)
    for blk in blocks:
        for name, fn, k in (("sad", lib.WelsHipPrimSampleSad, 1), ("satd", lib.WelsHipPrimSampleSatd, 1), ("sad_four", lib.WelsHipPrimSample4Sad, 4)):
            out, want = np.zeros(n * k, np.int32), np.zeros(n * k, np.int32)
            assert fn(blk, n, u8(p1), C.c_size_t(p1.size), s1, i32(o1), u8(p2), C.c_size_t(p2.size), s2, i32(o2), i32(out)) == 0
            getattr(orc, "orc_%s_batch" % name)(blk, n, u8(p1), s1, i32(o1), u8(p2), s2, i32(o2), i32(want))
            bad = np.nonzero(out != want)[0]
            assert bad.size == 0, (name, blk, int(bad[0]) // k, int(out[bad[0]]), int(want[bad[0]]))


def test_sad_satd_sad4(L):
    """131072 random pairs per block size and function, 7 x 3 x 131072 = 2.75 M comparisons, every case checked."""
    lib, orc = L
    rng = np.random.default_rng(11)
    p1, p2, o1, o2 = planes(rng, N_PAIRS)
    # (random u8 pixels: the domain of the reference's own unit tests; a tenth of the pairs with near-equal blocks, where SATD's rounding matters)
    p2[:, :128] = np.clip(p1.astype(np.int32) + rng.integers(-3, 4, p1.shape), 0, 255).astype(np.uint8)
    k = N_PAIRS // 10
    o2[:k] = (o1[:k] // 128) * 160 + o1[:k] % 128         # the same position in both planes: differences of -3 .. 3
    _sad_family(lib, orc, p1, 128, o1, p2, 160, o2)


def test_sad_satd_on_real_enc_ref_pairs(L, ref_tools, tmp_path):
    """Real (enc, ref) pairs as the motion search meets them, produced by oracle/_ref: enc = a macroblock of source picture k of the reference's own
    clip, ref = the reference encoder's RECONSTRUCTION of picture k - 1 (its bitstream through its decoder) displaced by a search offset -- every
    macroblock of every picture at 40 offsets ("real (enc, ref) MB pairs dumped from the oracle")."""
    import subprocess
    lib, orc = L
    assert ref_tools, "oracle/_ref must travel to the GPU box"
    w, h, frames = 320, 192, 8
    src = np.fromfile(os.path.join(ROOT, "oracle", "_ref", "res", "CiscoVT2people_320x192_12fps.yuv"), np.uint8)[: w * h * 3 // 2 * frames]
    fi, fo, fd = str(tmp_path / "in.yuv"), str(tmp_path / "ref.264"), str(tmp_path / "dec.yuv")
    src.tofile(fi)
    subprocess.check_call([ref_tools["enc"], "-i", fi, "-w", str(w), "-h", str(h), "-o", fo, "-rc", "-1", "-qp", "30", "-fps", "12", "-iper", "0", "-quiet"], stdout=subprocess.DEVNULL)
    subprocess.check_call([ref_tools["dec"], fo, fd], stdout=subprocess.DEVNULL)
    rec = np.fromfile(fd, np.uint8)
    fsz = w * h * 3 // 2
    assert rec.size == fsz * frames
    pad = 32                                              # the reference pictures' border, replicated as ExpandPicture does
    enc_y = np.concatenate([src[k * fsz:k * fsz + w * h] for k in range(1, frames)]).reshape((frames - 1) * h, w)
    ref_y = np.concatenate([np.pad(rec[k * fsz:k * fsz + w * h].reshape(h, w), pad, mode="edge") for k in range(frames - 1)])
    rs, rh = w + 2 * pad, h + 2 * pad
    rng = np.random.default_rng(17)
    o1, o2 = [], []
    for k in range(frames - 1):
        for my in range(h // 16):
            for mx in range(w // 16):
                d = rng.integers(-16, 17, (40, 2))
                d[0] = 0
                o1 += [(k * h + my * 16) * w + mx * 16] * 40
                o2 += [((k * rh + pad + my * 16 + int(dy)) * rs + pad + mx * 16 + int(dx)) for dx, dy in d]
    o1, o2 = np.array(o1, np.int32), np.array(o2, np.int32)
    assert len(o1) == 7 * 240 * 40
    _sad_family(lib, orc, np.ascontiguousarray(enc_y), w, o1, np.ascontiguousarray(ref_y), rs, o2)


def test_dct_quant_scan_dequant_idct(L):
    lib, orc = L
    rng = np.random.default_rng(12)
    n = N_PAIRS
    p1, p2, o1, o2 = planes(rng, n)
    dct, ref = np.zeros((n, 16), np.int16), np.zeros((n, 16), np.int16)
    assert lib.WelsHipPrimDctT4(n, u8(p1), C.c_size_t(p1.size), 128, i32(o1), u8(p2), C.c_size_t(p2.size), 160, i32(o2), i16(dct)) == 0
    orc.orc_dct4x4_batch(n, u8(p1), 128, i32(o1), u8(p2), 160, i32(o2), i16(ref))
    assert (dct == ref).all()
    for intra in (0, 1):
        x = rng.integers(-32768, 32768, (n, 16)).astype(np.int16)     # EncUT_EncoderMbAux.cpp:440-444 domain
        x[: n // 2] = dct[: n // 2]
        qp = rng.integers(0, 52, n).astype(np.uint8)
        io = x.copy()
        mx, zz, za = np.zeros(n, np.int16), np.zeros((n, 16), np.int16), np.zeros((n, 16), np.int16)
        ctr, nz = np.zeros(n, np.int32), np.zeros(n, np.int32)
        assert lib.WelsHipPrimQuant4x4(n, i16(io), u8(qp), intra, i16(mx), i16(zz), i16(za), i32(ctr), i32(nz)) == 0
        r = x.copy()
        rmx, rzz, rza = np.zeros(n, np.int16), np.zeros((n, 16), np.int16), np.zeros((n, 16), np.int16)
        rctr, rnz = np.zeros(n, np.int32), np.zeros(n, np.int32)
        orc.orc_quant_scan_batch(n, i16(r), u8(qp), intra, i16(rmx), i16(rzz), i16(rza), i32(rctr), i32(rnz))
        assert (io == r).all() and (mx == rmx).all() and (zz == rzz).all() and (za == rza).all() and (ctr == rctr).all() and (nz == rnz).all()
    lev = rng.integers(-2000, 2000, (n, 16)).astype(np.int16)
    lev[::4] = rng.integers(-32768, 32768, (len(lev[::4]), 16)).astype(np.int16)    # int16 wrap-around cases
    qp = rng.integers(0, 52, n).astype(np.uint8)
    pred = rng.integers(0, 256, (n, 16), dtype=np.uint8)
    rec, deq = np.zeros((n, 16), np.uint8), np.zeros((n, 16), np.int16)
    assert lib.WelsHipPrimDequantIDctRec(n, i16(lev), u8(qp), u8(pred), u8(rec), i16(deq)) == 0
    rrec, rdeq = np.zeros((n, 16), np.uint8), np.zeros((n, 16), np.int16)
    orc.orc_dequant_idct_rec_batch(n, i16(lev), u8(qp), u8(pred), u8(rrec), i16(rdeq))
    assert (deq == rdeq).all() and (rec == rrec).all()


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
        n = 4000
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
