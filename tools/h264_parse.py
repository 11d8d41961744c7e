#!/usr/bin/env python3
"""Minimal H.264 Baseline CAVLC syntax parser for the streams this encoder family writes (frame MBs only, POC type 2,
one reference, I and P slices).  Diagnostic tool: dumps one line per macroblock (type, prediction modes / motion vector
differences, cbp, qp delta, coefficient levels) so that two streams can be diffed at the syntax level:

    python tools/h264_parse.py a.264 > a.txt; python tools/h264_parse.py b.264 > b.txt; diff a.txt b.txt | head

The VLC tables are read from openh264_amd/csrc/common/h264_tables.h (entries are (bit_length << 8) | code).
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_tables():
    src = open(os.path.join(ROOT, "openh264_amd", "csrc", "common", "h264_tables.h")).read()
    out = {}
    for name in ("kWhCoeffToken", "kWhNcClass", "kWhTotalZeros", "kWhTotalZerosChromaDc", "kWhRunBefore", "kWhCbpCodeIntra", "kWhCbpCodeInter"):
        m = re.search(name + r"\[[^\]]*\]\s*=\s*\{([^}]*)\}", src)
        body = re.sub(r"//[^\n]*", "", m.group(1))
        out[name] = [int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", body)]
    return out


T = load_tables()


class Bits:
    def __init__(self, data):
        self.d, self.p = data, 0

    def u(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | ((self.d[self.p >> 3] >> (7 - (self.p & 7))) & 1)
            self.p += 1
        return v

    def peek(self, n):
        p = self.p
        v = 0
        for i in range(n):
            q = p + i
            b = (self.d[q >> 3] >> (7 - (q & 7))) & 1 if (q >> 3) < len(self.d) else 0
            v = (v << 1) | b
        return v

    def ue(self):
        z = 0
        while self.u(1) == 0:
            z += 1
        return (1 << z) - 1 + (self.u(z) if z else 0)

    def se(self):
        k = self.ue()
        return (k + 1) // 2 if k & 1 else -(k // 2)

    def more(self):
        # more_rbsp_data: anything but the trailing 1000... left
        last = len(self.d) * 8 - 1
        while last >= 0 and not (self.d[last >> 3] >> (7 - (last & 7))) & 1:
            last -= 1
        return self.p < last


def vlc_lookup(bits, table_slice):
    """table_slice: list of (len<<8|code); returns the index whose code matches the next bits."""
    for idx, e in enumerate(table_slice):
        ln, code = e >> 8, e & 0xff
        if ln and bits.peek(ln) == code:
            bits.p += ln
            return idx
    raise ValueError("no VLC match at bit %d" % bits.p)


def residual_block(bits, nc, max_coeff):
    """Returns (total_coeff, list of levels in zig-zag order, length max_coeff)."""
    cls = T["kWhNcClass"][nc]
    base = cls * 17 * 4
    idx = vlc_lookup(bits, T["kWhCoeffToken"][base:base + 68])
    total, t1 = idx >> 2, idx & 3
    coef = [0] * max_coeff
    if total == 0:
        return 0, coef
    levels = []
    suffix_len = 1 if (total > 10 and t1 < 3) else 0
    for k in range(total):
        if k < t1:
            levels.append(-1 if bits.u(1) else 1)
            continue
        prefix = 0
        while bits.u(1) == 0:
            prefix += 1
        code = min(15, prefix) << suffix_len
        if suffix_len > 0 or prefix >= 14:
            size = suffix_len
            if prefix == 14 and suffix_len == 0:
                size = 4
            if prefix >= 15:
                size = prefix - 3
            code += bits.u(size) if size else 0
        if prefix >= 15 and suffix_len == 0:
            code += 15
        if prefix >= 16:
            code += (1 << (prefix - 3)) - 4096
        if k == t1 and t1 < 3:
            code += 2
        val = (code + 2) >> 1 if code % 2 == 0 else -((code + 1) >> 1)
        levels.append(val)
        if suffix_len == 0:
            suffix_len = 1
        if abs(val) > (3 << (suffix_len - 1)) and suffix_len < 6:
            suffix_len += 1
    zeros_left = 0
    if total < max_coeff:
        if nc == 17:
            zeros_left = vlc_lookup(bits, T["kWhTotalZerosChromaDc"][total * 4:total * 4 + 4])
        else:
            zeros_left = vlc_lookup(bits, T["kWhTotalZeros"][total * 16:total * 16 + 16])
    runs = []
    for k in range(total - 1):
        if zeros_left > 0:
            zl = min(zeros_left, 7)
            r = vlc_lookup(bits, T["kWhRunBefore"][zl * 15:zl * 15 + 15])
        else:
            r = 0
        runs.append(r)
        zeros_left -= r
    runs.append(zeros_left)
    pos = -1
    for k in range(total - 1, -1, -1):
        pos += runs[k] + 1
        coef[pos] = levels[k]
    return total, coef


def blk_raster(b):
    return (((b >> 1) & 1) | ((b >> 2) & 2)) * 4 + ((b & 1) | ((b >> 1) & 2))


def nc_of(na, nb):
    if na >= 0 and nb >= 0:
        return (na + nb + 1) >> 1
    if na >= 0:
        return na
    if nb >= 0:
        return nb
    return 0


def parse_stream(data, out=sys.stdout, levels=True):
    nal_pos = [m.start() for m in re.finditer(b"\x00\x00\x01", data)]
    sps = {}
    pic = 0
    nzc_pic = None
    for k, st in enumerate(nal_pos):
        end = nal_pos[k + 1] if k + 1 < len(nal_pos) else len(data)
        nal = data[st + 3:end]
        while nal and nal[-1] == 0 and k + 1 < len(nal_pos):
            nal = nal[:-1]
        hdr = nal[0]
        rbsp = re.sub(b"\x00\x00\x03", b"\x00\x00", nal[1:])
        typ = hdr & 31
        b = Bits(rbsp)
        if typ == 7:
            b.u(24)
            b.ue()
            sps["log2_fn"] = b.ue() + 4
            b.ue(); b.ue(); b.u(1)
            sps["mb_w"] = b.ue() + 1
            sps["mb_h"] = b.ue() + 1
            print("SPS %dx%d MBs" % (sps["mb_w"], sps["mb_h"]), file=out)
        elif typ == 8:
            print("PPS", file=out)
        elif typ in (1, 5):
            mb_w, mb_h = sps["mb_w"], sps["mb_h"]
            first = b.ue()
            stype = b.ue() % 5
            b.ue()
            b.u(sps["log2_fn"])
            if typ == 5:
                b.ue()
            if stype == 0:
                if b.u(1):
                    b.ue()
                if b.u(1):                     # ref_pic_list_modification
                    while True:
                        idc = b.ue()
                        if idc == 3:
                            break
                        b.ue()
            if hdr >> 5:
                if typ == 5:
                    b.u(2)
                elif b.u(1):                   # adaptive_ref_pic_marking_mode_flag: memory_management_control_operations
                    while True:
                        op = b.ue()
                        if op == 0:
                            break
                        if op in (1, 3):
                            b.ue()
                        if op == 2:
                            b.ue()
                        if op in (3, 6):
                            b.ue()
                        if op == 4:
                            b.ue()
            qp = 26 + b.se()
            idc = b.ue()
            if idc != 1:
                b.se(); b.se()
            if first == 0:
                pic += 1
                nzc_pic = [[-1] * 24 for _ in range(mb_w * mb_h)]
                slice_of = [-1] * (mb_w * mb_h)
                slice_no = 0
            else:
                slice_no += 1
            print("SLICE pic %d type %s first_mb %d qp %d idc %d" % (pic, "PI"[stype == 2], first, qp, idc), file=out)
            xy = first
            skip_run_pending = 0
            while True:
                if stype == 0:
                    run = b.ue()
                    for _ in range(run):
                        slice_of[xy] = slice_no
                        nzc_pic[xy] = [0] * 24
                        print("  mb %4d P_Skip" % xy, file=out)
                        xy += 1
                    if not b.more():
                        break
                mbt = b.ue()
                desc = ""
                intra = stype == 2 or mbt >= 5
                it = mbt - (5 if stype == 0 else 0)
                i16 = False
                if intra:
                    if it == 0:
                        modes = []
                        for _ in range(16):
                            modes.append(-1 if b.u(1) else b.u(3))
                        desc = "I4x4 modes %s chroma %d" % (modes, b.ue())
                    else:
                        i16 = True
                        it -= 1
                        cbp16 = (15 if it >= 12 else 0) | (((it % 12) >> 2) << 4)
                        desc = "I16x16 mode %d chroma %d" % (it & 3, b.ue())
                else:
                    if mbt == 0:
                        desc = "P16x16 mvd %s" % [(b.se(), b.se())]
                    elif mbt in (1, 2):
                        desc = "%s mvd %s" % ("P16x8" if mbt == 1 else "P8x16", [(b.se(), b.se()), (b.se(), b.se())])
                    else:
                        subs = [b.ue() for _ in range(4)]
                        if mbt == 3:
                            raise ValueError("P8x8 with ref_idx not handled")
                        mv = []
                        for s in subs:
                            for _ in range((1, 2, 2, 4)[s]):
                                mv.append((b.se(), b.se()))
                        desc = "P8x8 sub %s mvd %s" % (subs, mv)
                if i16:
                    cbp = cbp16
                else:
                    code = b.ue()
                    tab = T["kWhCbpCodeIntra"] if intra else T["kWhCbpCodeInter"]
                    cbp = tab.index(code)
                slice_of[xy] = slice_no
                nz = [0] * 24
                nzc_pic[xy] = nz
                lv_txt = ""
                if cbp > 0 or i16:
                    dqp = b.se()
                    desc += " cbp %d dqp %d" % (cbp, dqp)
                    mbx = xy % mb_w
                    left = nzc_pic[xy - 1] if mbx > 0 and slice_of[xy - 1] == slice_no else None
                    top = nzc_pic[xy - mb_w] if xy >= mb_w and slice_of[xy - mb_w] == slice_no else None

                    def la(r):
                        return nz[r - 1] if r & 3 else (left[r + 3] if left else -1)

                    def lb(r):
                        return nz[r - 4] if r >> 2 else (top[r + 12] if top else -1)
                    blocks = []
                    if i16:
                        _, c = residual_block(b, nc_of(la(0), lb(0)), 16)
                        blocks.append(("dc", c))
                    for blk in range(16):
                        if not (cbp & (1 << (blk >> 2))):
                            continue
                        r = blk_raster(blk)
                        tc, c = residual_block(b, nc_of(la(r), lb(r)), 15 if i16 else 16)
                        nz[r] = tc
                        blocks.append(("y%d" % blk, c))
                    if cbp >> 4:
                        for p in range(2):
                            _, c = residual_block(b, 17, 4)
                            blocks.append(("cdc%d" % p, c))
                        if (cbp >> 4) & 2:
                            for p in range(2):
                                for cidx in range(4):
                                    o = 16 + p * 4
                                    a = nz[o + cidx - 1] if cidx & 1 else (left[o + cidx + 1] if left else -1)
                                    bb = nz[o + cidx - 2] if cidx >> 1 else (top[o + cidx + 2] if top else -1)
                                    tc, c = residual_block(b, nc_of(a, bb), 15)
                                    nz[o + cidx] = tc
                                    blocks.append(("c%d" % (p * 4 + cidx), c))
                    if levels:
                        lv_txt = "".join("\n      %-5s %s" % (n, c) for n, c in blocks)
                else:
                    desc += " cbp 0"
                print("  mb %4d bit %6d %s%s" % (xy, b.p, desc, lv_txt), file=out)
                xy += 1
                if not b.more():
                    break


if __name__ == "__main__":
    parse_stream(open(sys.argv[1], "rb").read(), levels="--no-levels" not in sys.argv)
