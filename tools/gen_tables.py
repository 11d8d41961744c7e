#!/usr/bin/env python3
"""Generate openh264_amd/csrc/common/h264_tables.h.

All numbers in that header are *constants of the H.264 standard* (ITU-T H.264: Tables 8-15..8-17,
9-4, 9-5, 9-7..9-10, A-1) or fixed-point design constants of the reference encoder that the
bitstream parity contract depends on (quantiser MF/FF triples, lambda-per-QP).  To avoid
transcription errors the ones the built reference exports as data symbols are read back from
oracle/_ref/libref_openh264.so (built by oracle/Makefile) and re-laid-out in our own compact form;
the rest are typed from the standard here and cross-checked in tests/test_tables.py.

Run:  python tools/gen_tables.py   (needs oracle/_ref; the generated header is committed)
"""
import ctypes, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libref_openh264.so")
OUT = os.path.join(ROOT, "openh264_amd", "csrc", "common", "h264_tables.h")


def arr(lib, sym, ctype, n):
    return list((ctype * n).in_dll(lib, sym))


def cls3(row8):
    # 4x4 position class pattern {a,b,a,b,b,c,b,c}: a=(even,even) b=mixed c=(odd,odd)
    a, b, c = row8[0], row8[1], row8[5]
    assert list(row8) == [a, b, a, b, b, c, b, c], row8
    return a, b, c


def fmt(vals, per=16, w=4):
    lines = []
    for i in range(0, len(vals), per):
        lines.append("  " + ",".join(f"{v:>{w}}" for v in vals[i:i + per]) + ",")
    return "\n".join(lines)


# ---- Intra4x4 predictors as a table look-up (kernels/intra_mb.h kWhI4Desc) -------------------------------------------
# Edge samples E(0..12) = L3 L2 L1 L0 TL T0..T7; X = E(0), E(0) .. E(12), E(12); RAW[a] = X[a], F2[a] = (X[a] + X[a+1] + 1) >> 1,
# F3[a] = (X[a] + 2 X[a+1] + X[a+2] + 2) >> 2.  i4_desc gives, for a sample of a mode, the table offset (RAW 0.., F2 16.., F3 32.., DC 48);
# i4_ref is the predictor as kernels/intra_mb.h wh_pred4_px (= the reference's WelsI4x4Luma*Pred_c, get_intra_predictor.cpp) computes it.
def i4_desc(mode, x, y):
    RAW, F2, F3 = 0, 16, 32
    if mode == 0: return RAW + 6 + x
    if mode == 1: return RAW + 4 - y
    if mode == 2: return 48
    if mode == 3: return F3 + 12 if (x == 3 and y == 3) else F3 + 6 + x + y
    if mode == 4: return F3 + 4 + x - y
    if mode == 5:
        z, i = 2 * x - y, x - (y >> 1)
        if z >= 0: return (F3 + 4 + i) if (z & 1) else (F2 + 5 + i)
        return F3 + 4 if z == -1 else F3 + 5 - y
    if mode == 6:
        z, j = 2 * y - x, y - (x >> 1)
        if z >= 0: return (F3 + 4 - j) if (z & 1) else (F2 + 4 - j)
        return F3 + 4 if z == -1 else F3 + 3 + x
    if mode == 7:
        i = x + (y >> 1)
        return (F3 + 6 + i) if (y & 1) else (F2 + 6 + i)
    z, j = x + 2 * y, y + (x >> 1)
    if z > 5: return RAW + 1
    if z == 5: return F3 + 0
    return (F3 + 2 - j) if (z & 1) else (F2 + 3 - j)


def i4_ref(mode, x, y, e, dc):
    f3 = lambda a, b, c: (a + 2 * b + c + 2) >> 2
    f2 = lambda a, b: (a + b + 1) >> 1
    if mode == 0: return e[5 + x]
    if mode == 1: return e[3 - y]
    if mode == 2: return dc
    if mode == 3:
        return (e[11] + 3 * e[12] + 2) >> 2 if (x == 3 and y == 3) else f3(e[5 + x + y], e[6 + x + y], e[7 + x + y])
    if mode == 4: return f3(e[4 + x - y - 1], e[4 + x - y], e[4 + x - y + 1])
    if mode == 5:
        z, i = 2 * x - y, x - (y >> 1)
        if z >= 0: return f3(e[5 + i - 2], e[5 + i - 1], e[5 + i]) if z & 1 else f2(e[5 + i - 1], e[5 + i])
        return f3(e[3], e[4], e[5]) if z == -1 else f3(e[3 - (y - 1)], e[3 - (y - 2)], e[3 - (y - 3)])
    if mode == 6:
        z, j = 2 * y - x, y - (x >> 1)
        if z >= 0: return f3(e[3 - (j - 2)], e[3 - (j - 1)], e[3 - j]) if z & 1 else f2(e[3 - (j - 1)], e[3 - j])
        return f3(e[3], e[4], e[5]) if z == -1 else f3(e[5 + x - 1], e[5 + x - 2], e[5 + x - 3])
    if mode == 7:
        i = x + (y >> 1)
        return f3(e[5 + i], e[5 + i + 1], e[5 + i + 2]) if y & 1 else f2(e[5 + i], e[5 + i + 1])
    z, j = x + 2 * y, y + (x >> 1)
    if z > 5: return e[0]
    if z == 5: return (e[1] + 3 * e[0] + 2) >> 2
    return f3(e[3 - j], e[3 - (j + 1)], e[3 - (j + 2)]) if z & 1 else f2(e[3 - j], e[3 - (j + 1)])


def i4_table(check=2000):
    """The 36 words of kWhI4Desc ([mode * 4 + row], byte x = offset of sample (x, row)), checked against i4_ref on random edges."""
    import random
    rnd = random.Random(4)
    for _ in range(check):
        e = [rnd.randint(0, 255) for _ in range(13)]
        X = [e[0]] + e + [e[12]] * 3
        T = [0] * 49
        for a in range(15):
            T[a], T[16 + a], T[32 + a] = X[a], (X[a] + X[a + 1] + 1) >> 1, (X[a] + 2 * X[a + 1] + X[a + 2] + 2) >> 2
        T[48] = rnd.randint(0, 255)
        for m in range(9):
            for y in range(4):
                for x in range(4):
                    assert T[i4_desc(m, x, y)] == i4_ref(m, x, y, e, T[48]), (m, x, y)
    return [sum(i4_desc(m, x, r) << (8 * x) for x in range(4)) for m in range(9) for r in range(4)]


def main():
    if "--i4" in sys.argv:
        print(", ".join("0x%08xu" % w for w in i4_table()))
        return
    lib = ctypes.CDLL(LIB)
    mf = arr(lib, "_ZN7WelsEnc11g_kiQuantMFE", ctypes.c_int16, 52 * 8)
    ff = arr(lib, "_ZN7WelsEnc16g_kiQuantInterFFE", ctypes.c_int16, 58 * 8)
    dq = arr(lib, "_ZN10WelsCommon17g_kuiDequantCoeffE", ctypes.c_uint16, 52 * 8)
    cqp = arr(lib, "_ZN10WelsCommon18g_kuiChromaQpTableE", ctypes.c_uint8, 52)
    lam = arr(lib, "_ZN7WelsEnc15g_kiQpCostTableE", ctypes.c_int32, 52)
    ctok = arr(lib, "_ZN7WelsEnc18g_kuiVlcCoeffTokenE", ctypes.c_uint8, 5 * 17 * 4 * 2)
    tz = arr(lib, "_ZN7WelsEnc18g_kuiVlcTotalZerosE", ctypes.c_uint8, 16 * 16 * 2)
    tzc = arr(lib, "_ZN7WelsEnc26g_kuiVlcTotalZerosChromaDcE", ctypes.c_uint8, 4 * 4 * 2)
    rb = arr(lib, "_ZN7WelsEnc17g_kuiVlcRunBeforeE", ctypes.c_uint8, 8 * 15 * 2)
    ncmap = arr(lib, "_ZN7WelsEnc18g_kuiEncNcMapTableE", ctypes.c_uint8, 18)

    mf3 = [v for q in range(52) for v in cls3(mf[q * 8:q * 8 + 8])]
    ff3 = [v for q in range(58) for v in cls3(ff[q * 8:q * 8 + 8])]
    dq3 = [v for q in range(52) for v in cls3(dq[q * 8:q * 8 + 8])]

    # ---- typed from the standard ------------------------------------------------------------
    # Table 9-4: codeNum -> coded_block_pattern (Intra_4x4 / Inter), ChromaArrayType 1
    cbp_intra_by_code = [47, 31, 15, 0, 23, 27, 29, 30, 7, 11, 13, 14, 39, 43, 45, 46, 16, 3, 5, 10, 12, 19, 21, 26,
                         28, 35, 37, 42, 44, 1, 2, 4, 8, 17, 18, 20, 24, 6, 9, 22, 25, 32, 33, 34, 36, 40, 38, 41]
    cbp_inter_by_code = [0, 16, 1, 2, 4, 8, 32, 3, 5, 10, 12, 15, 47, 7, 11, 13, 14, 6, 9, 31, 35, 37, 42, 44,
                         33, 34, 36, 40, 39, 43, 45, 46, 17, 18, 20, 24, 19, 21, 26, 28, 23, 27, 29, 30, 22, 25, 38, 41]
    code_intra = [cbp_intra_by_code.index(c) for c in range(48)]
    code_inter = [cbp_inter_by_code.index(c) for c in range(48)]
    # Table 8-16 alpha', beta' for indexA/indexB 0..51
    alpha = [0] * 16 + [4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71,
                        80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255]
    beta = [0] * 16 + [2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12,
                       13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18]
    # Table 8-17 tC0 for bS = 1,2,3
    tc0 = [(0, 0, 0)] * 17 + [(0, 0, 1)] * 4 + [(0, 1, 1)] * 2 + [(1, 1, 1)] * 4 + [(1, 1, 2)] * 4 + \
          [(1, 2, 3), (1, 2, 3), (2, 2, 3), (2, 2, 4), (2, 3, 4), (2, 3, 4), (3, 3, 5), (3, 4, 6), (3, 4, 6),
           (4, 5, 7), (4, 5, 8), (4, 6, 9), (5, 7, 10), (6, 8, 11), (6, 8, 13), (7, 10, 14), (8, 11, 16),
           (9, 12, 18), (10, 13, 20), (11, 15, 23), (13, 17, 25)]
    assert len(alpha) == 52 and len(beta) == 52 and len(tc0) == 52, (len(alpha), len(beta), len(tc0))
    # Table A-1 level limits: (level_idc, MaxMBPS, MaxFS, MaxDpbMbs, MaxBR, MaxCPB); level 1b carried as idc 9
    levels = [(10, 1485, 99, 396, 64, 175), (9, 1485, 99, 396, 128, 350), (11, 3000, 396, 900, 192, 500),
              (12, 6000, 396, 2376, 384, 1000), (13, 11880, 396, 2376, 768, 2000), (20, 11880, 396, 2376, 2000, 2000),
              (21, 19800, 792, 4752, 4000, 4000), (22, 20250, 1620, 8100, 4000, 4000),
              (30, 40500, 1620, 8100, 10000, 10000), (31, 108000, 3600, 18000, 14000, 14000),
              (32, 216000, 5120, 20480, 20000, 20000), (40, 245760, 8192, 32768, 20000, 25000),
              (41, 245760, 8192, 32768, 50000, 62500), (42, 522240, 8704, 34816, 50000, 62500),
              (50, 589824, 22080, 110400, 135000, 135000), (51, 983040, 36864, 184320, 240000, 240000),
              (52, 2073600, 36864, 184320, 240000, 240000)]

    o = []
    o.append("// GENERATED by tools/gen_tables.py -- do not edit.  H.264 standard constants + the fixed-point\n"
             "// quantiser design constants the bitstream parity contract depends on (see the generator).\n"
             "#pragma once\n#include <stdint.h>\n\n#ifndef WH_TABLE\n#define WH_TABLE static const\n#endif\n")
    o.append("// Position class of coefficient i (raster 4x4): 0=(even,even) 1=mixed 2=(odd,odd)\n"
             "#define WH_POSCLASS(i) ((((i) >> 2) & 1) + ((i) & 1))\n")
    o.append("// forward quant multiplier, >>16 form, [qp][class]\nWH_TABLE int32_t kWhQuantMF[52 * 3] = {\n" + fmt(mf3, 12, 6) + "\n};")
    o.append("// forward quant rounding offset: inter rows 0..51, intra = row qp+6 [qp][class]\nWH_TABLE int32_t kWhQuantFF[58 * 3] = {\n" + fmt(ff3, 12, 4) + "\n};")
    o.append("// dequant scale LevelScale(qp%6)<<(qp/6) [qp][class]\nWH_TABLE int32_t kWhDequant[52 * 3] = {\n" + fmt(dq3, 12, 5) + "\n};")
    o.append("// QPc as a function of qPI (Table 8-15)\nWH_TABLE int32_t kWhChromaQp[52] = {\n" + fmt(cqp, 26, 2) + "\n};")
    o.append("// mode-decision lambda per QP\nWH_TABLE int32_t kWhLambda[52] = {\n" + fmt(lam, 26, 2) + "\n};")
    o.append("// deblocking alpha'/beta' (Table 8-16), tC0 for bS 1..3 (Table 8-17)\nWH_TABLE int32_t kWhAlpha[52] = {\n" + fmt(alpha, 26, 3) + "\n};")
    o.append("WH_TABLE int32_t kWhBeta[52] = {\n" + fmt(beta, 26, 2) + "\n};")
    o.append("WH_TABLE uint8_t kWhTc0[52 * 3] = {\n" + fmt([v for t in tc0 for v in t], 24, 2) + "\n};")
    o.append("// the same, one 32-bit word per indexA: tc0 for bS 1 | bS 2 << 8 | bS 3 << 16 (a wave-uniform scalar load)\nWH_TABLE int32_t kWhTc0Packed[52] = {\n" + fmt([t[0] | (t[1] << 8) | (t[2] << 16) for t in tc0], 13, 8) + "\n};")
    o.append("// me(v) codeNum for coded_block_pattern (Table 9-4), indexed by cbp\nWH_TABLE uint8_t kWhCbpCodeIntra[48] = {\n" + fmt(code_intra, 16, 2) + "\n};")
    o.append("WH_TABLE uint8_t kWhCbpCodeInter[48] = {\n" + fmt(code_inter, 16, 2) + "\n};")
    # VLC tables: pack as (len<<8)|code in uint16
    def pack(lst):
        return [(lst[2 * i + 1] << 8) | lst[2 * i] for i in range(len(lst) // 2)]
    o.append("// CAVLC tables, entries are (bit_length<<8)|code_value\n"
             "// coeff_token (Table 9-5) [nC class 0..4][TotalCoeff 0..16][TrailingOnes 0..3]; class 4 = ChromaDC\n"
             "WH_TABLE uint16_t kWhCoeffToken[5 * 17 * 4] = {\n" + fmt(pack(ctok), 8, 6) + "\n};")
    o.append("// nC (0..16, index 17 = ChromaDC) -> coeff_token table class\nWH_TABLE uint8_t kWhNcClass[18] = {\n" + fmt(ncmap, 18, 1) + "\n};")
    o.append("// total_zeros (Tables 9-7, 9-8) [TotalCoeff][total_zeros]\nWH_TABLE uint16_t kWhTotalZeros[16 * 16] = {\n" + fmt(pack(tz), 16, 5) + "\n};")
    o.append("// total_zeros for chroma DC 2x2 (Table 9-9a) [TotalCoeff][total_zeros]\nWH_TABLE uint16_t kWhTotalZerosChromaDc[4 * 4] = {\n" + fmt(pack(tzc), 4, 5) + "\n};")
    o.append("// run_before (Table 9-10) [min(zerosLeft,7)][run_before]\nWH_TABLE uint16_t kWhRunBefore[8 * 15] = {\n" + fmt(pack(rb), 15, 5) + "\n};")
    o.append("// Table A-1: level_idc (9 = 1b), MaxMBPS, MaxFS, MaxDpbMbs, MaxBR, MaxCPB\nWH_TABLE uint32_t kWhLevelLimits[17 * 6] = {\n" + fmt([v for l in levels for v in l], 6, 8) + "\n};")
    with open(OUT, "w") as f:
        f.write("\n".join(o) + "\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
