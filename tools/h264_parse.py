#!/usr/bin/env python3
"""Minimal Annex-B H.264 header parser (SPS / PPS / slice header) used by tests and during
development to inspect what an encoder wrote.  Baseline/Main syntax without SVC extensions."""
import sys


class BitReader:
    def __init__(self, data):
        self.d = data
        self.p = 0

    def u(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | ((self.d[self.p >> 3] >> (7 - (self.p & 7))) & 1)
            self.p += 1
        return v

    def ue(self):
        z = 0
        while self.u(1) == 0:
            z += 1
        return (1 << z) - 1 + (self.u(z) if z else 0)

    def se(self):
        k = self.ue()
        return (k + 1) // 2 if k & 1 else -(k // 2)


def split_nals(data):
    out = []
    i = 0
    n = len(data)
    starts = []
    while i + 3 <= n:
        if data[i] == 0 and data[i + 1] == 0 and data[i + 2] == 1:
            starts.append((i + 3, i - 1 if i > 0 and data[i - 1] == 0 else i))
            i += 3
        else:
            i += 1
    for k, (s, sc) in enumerate(starts):
        e = starts[k + 1][1] if k + 1 < len(starts) else n
        out.append(data[s:e])
    return out


def unescape(nal):
    out = bytearray()
    z = 0
    for b in nal:
        if z >= 2 and b == 3:
            z = 0
            continue
        out.append(b)
        z = z + 1 if b == 0 else 0
    return bytes(out)


def parse_sps(r):
    s = {}
    s['profile_idc'] = r.u(8)
    s['constraint'] = r.u(8)
    s['level_idc'] = r.u(8)
    s['sps_id'] = r.ue()
    s['log2_max_frame_num'] = r.ue() + 4
    s['poc_type'] = r.ue()
    if s['poc_type'] == 0:
        s['log2_max_poc_lsb'] = r.ue() + 4
    s['num_ref_frames'] = r.ue()
    s['gaps'] = r.u(1)
    s['mb_w'] = r.ue() + 1
    s['mb_h'] = r.ue() + 1
    s['frame_mbs_only'] = r.u(1)
    s['direct8x8'] = r.u(1)
    s['crop'] = r.u(1)
    if s['crop']:
        s['crop_lrtb'] = [r.ue() for _ in range(4)]
    s['vui'] = r.u(1)
    if s['vui']:
        v = {}
        v['aspect'] = r.u(1)
        if v['aspect']:
            idc = r.u(8)
            v['aspect_idc'] = idc
            if idc == 255:
                r.u(32)
        v['overscan'] = r.u(1)
        v['video_signal'] = r.u(1)
        if v['video_signal']:
            r.u(3); r.u(1)
            if r.u(1):
                r.u(24)
        v['chroma_loc'] = r.u(1)
        v['timing'] = r.u(1)
        v['nal_hrd'] = r.u(1)
        v['vcl_hrd'] = r.u(1)
        v['pic_struct'] = r.u(1)
        v['bs_restriction'] = r.u(1)
        if v['bs_restriction']:
            v['mv_over_pic'] = r.u(1)
            v['rest'] = [r.ue() for _ in range(6)]
        s['vui_params'] = v
    return s


def parse_pps(r):
    p = {}
    p['pps_id'] = r.ue(); p['sps_id'] = r.ue(); p['cabac'] = r.u(1); p['pic_order_present'] = r.u(1)
    p['slice_groups'] = r.ue() + 1
    p['num_ref_l0'] = r.ue() + 1; p['num_ref_l1'] = r.ue() + 1
    p['weighted'] = r.u(1); p['weighted_bipred'] = r.u(2)
    p['init_qp'] = r.se() + 26; p['init_qs'] = r.se() + 26; p['chroma_qp_off'] = r.se()
    p['deblock_ctrl'] = r.u(1); p['constrained_intra'] = r.u(1); p['redundant'] = r.u(1)
    return p


def parse_slice(r, nal_type, nal_ref_idc, sps, pps):
    h = {}
    h['first_mb'] = r.ue(); h['slice_type'] = r.ue(); h['pps_id'] = r.ue()
    h['frame_num'] = r.u(sps['log2_max_frame_num'])
    if nal_type == 5:
        h['idr_pic_id'] = r.ue()
    if sps['poc_type'] == 0:
        h['poc_lsb'] = r.u(sps['log2_max_poc_lsb'])
    st = h['slice_type'] % 5
    if st == 0:
        h['num_ref_override'] = r.u(1)
        if h['num_ref_override']:
            h['num_ref_l0'] = r.ue() + 1
        h['reorder_flag'] = r.u(1)
        if h['reorder_flag']:
            ops = []
            while True:
                idc = r.ue()
                if idc == 3:
                    break
                ops.append((idc, r.ue()))
            h['reorder'] = ops
    if nal_ref_idc:
        if nal_type == 5:
            h['no_output_prior'] = r.u(1); h['long_term_ref'] = r.u(1)
        else:
            h['adaptive_marking'] = r.u(1)
            if h['adaptive_marking']:
                ops = []
                while True:
                    op = r.ue()
                    if op == 0:
                        break
                    a = [op]
                    if op in (1, 3): a.append(r.ue())
                    if op == 2: a.append(r.ue())
                    if op in (3, 6): a.append(r.ue())
                    if op == 4: a.append(r.ue())
                    ops.append(tuple(a))
                h['mmco'] = ops
    if pps['cabac'] and st != 2:
        h['cabac_init_idc'] = r.ue()
    h['slice_qp_delta'] = r.se()
    if pps['deblock_ctrl']:
        h['disable_deblock_idc'] = r.ue()
        if h['disable_deblock_idc'] != 1:
            h['alpha_div2'] = r.se(); h['beta_div2'] = r.se()
    h['header_bits'] = r.p
    return h


def dump(path, max_nals=20):
    data = open(path, 'rb').read()
    spss, ppss = {}, {}
    for i, nal in enumerate(split_nals(data)):
        if i >= max_nals:
            break
        rb = unescape(nal)
        t = rb[0] & 31
        ref = (rb[0] >> 5) & 3
        r = BitReader(rb[1:] + b'\0\0\0\0')
        if t == 7:
            s = parse_sps(r); spss[s['sps_id']] = s
            print(i, 'SPS', len(nal), s)
        elif t == 8:
            p = parse_pps(r); ppss[p['pps_id']] = p
            print(i, 'PPS', len(nal), p)
        elif t in (1, 5):
            # need pps id: peek
            r2 = BitReader(rb[1:] + b'\0\0\0\0'); r2.ue(); r2.ue(); pid = r2.ue()
            pps = ppss[pid]; sps = spss[pps['sps_id']]
            print(i, 'SLICE nal_type', t, 'ref_idc', ref, len(nal), parse_slice(r, t, ref, sps, pps))
        else:
            print(i, 'NAL type', t, len(nal))


if __name__ == '__main__':
    dump(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20)
