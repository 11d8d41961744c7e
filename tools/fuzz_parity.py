#!/usr/bin/env python3
"""Randomised bitstream parity: kernel sources (CPU wave-emulation build, or libwelship.so with --hip) against
oracle/_ref/ref_enc run live, over random picture sizes, QPs, slice counts, complexity modes, deblocking modes
and content classes.  Test infrastructure: prints one line per case and a summary; exit code 1 on any mismatch.

    python tools/fuzz_parity.py --cases 200 --seed 1            # CPU emulation
    python tools/fuzz_parity.py --cases 40 --hip                # on an MI355X box
"""
import argparse
import ctypes as C
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import openh264_amd as oh                                  # noqa: E402
from openh264_amd import build as B                        # noqa: E402
from openh264_amd.utils.synth import synth_sequence        # noqa: E402


def rng2_odd(w, h):
    """Every ninth size or so becomes odd in one or both dimensions (decided from the size itself: no draw)."""
    k = (w * 7 + h * 13) % 19
    if w < 18 or h < 18 or k > 2:
        return None
    return ((1, 0), (0, 1), (1, 1))[k]


def content(kind, w, h, frames, rng):
    """I420 bytes of one of several content classes (all integer, seeded)."""
    if kind == "oddsize":                                  # smooth-ish noise, any picture size
        fsz = w * h * 3 // 2
        out = bytearray()
        base = rng.integers(0, 256, fsz, dtype=np.uint8).astype(np.int32)
        for n in range(frames):
            v = (base * 3 + np.roll(base, 1) + np.roll(base, w) * 2 + np.roll(base, 3 * n + 1) * 2 + 4) >> 3
            out += np.clip(v + rng.integers(-3, 4, fsz), 0, 255).astype(np.uint8).tobytes()
        return bytes(out)
    if kind == "synth":
        return synth_sequence(w, h, frames, seed=int(rng.integers(1, 1 << 30)))
    out = bytearray()
    yy, xx = np.mgrid[0:h, 0:w]
    base = rng.integers(0, 256, (h, w)).astype(np.int32)
    if kind == "noise":                                    # uncorrelated noise, new every frame
        for _ in range(frames):
            out += rng.integers(0, 256, h * w, dtype=np.uint8).tobytes()
            out += rng.integers(0, 256, h * w // 2, dtype=np.uint8).tobytes()
        return bytes(out)
    if kind == "flat":                                     # constant planes with a few outliers (skip-heavy)
        y0 = np.full((h, w), int(rng.integers(0, 256)), np.int32)
        for n in range(frames):
            y = y0.copy()
            for _ in range(3):
                py, px = int(rng.integers(0, h)), int(rng.integers(0, w))
                y[py:py + 5, px:px + 7] = int(rng.integers(0, 256))
            out += y.astype(np.uint8).tobytes()
            out += np.full(h * w // 4, int(rng.integers(0, 256)), np.uint8).tobytes() * 2
        return bytes(out)
    if kind == "extreme":                                  # saturated checker patterns: clipping paths, large levels
        p = int(rng.integers(1, 9))
        for n in range(frames):
            y = ((((xx + n * 3) // p + (yy + n) // p) & 1) * 255).astype(np.uint8)
            c = ((((xx[:h // 2, :w // 2] + n) // p) & 1) * 255).astype(np.uint8)
            out += y.tobytes() + c.tobytes() + (255 - c).tobytes()
        return bytes(out)
    if kind == "fastmotion":                               # large global motion (beyond the search range) + texture
        acc = np.zeros_like(base)
        for d in range(4):
            acc += np.roll(base, d, axis=1) + np.roll(base, d, axis=0)
        tex = (acc >> 3)
        dx, dy = int(rng.integers(-40, 41)), int(rng.integers(-24, 25))
        for n in range(frames):
            y = np.roll(np.roll(tex, dy * n, axis=0), dx * n, axis=1)
            out += np.clip(y, 0, 255).astype(np.uint8).tobytes()
            cu = np.roll(tex[::2, ::2], dx * n // 2, axis=1)
            out += np.clip(cu, 0, 255).astype(np.uint8).tobytes()
            out += np.clip(255 - cu, 0, 255).astype(np.uint8).tobytes()
        return bytes(out)
    if kind == "subpel":                                   # smooth gradients moving by fractions of a pixel
        fx, fy = rng.integers(1, 8, 2)
        for n in range(frames):
            y = (128 + 100 * np.sin((xx * 4 + n * fx) / 37.0) * np.cos((yy * 4 + n * fy) / 29.0)).astype(np.int32)
            y += rng.integers(-2, 3, (h, w))
            out += np.clip(y, 0, 255).astype(np.uint8).tobytes()
            c = (128 + 60 * np.sin((xx[:h // 2, :w // 2] * 8 + n * fx) / 41.0)).astype(np.int32)
            out += np.clip(c, 0, 255).astype(np.uint8).tobytes()
            out += np.clip(255 - c, 0, 255).astype(np.uint8).tobytes()
        return bytes(out)
    raise ValueError(kind)


LEVEL1 = False
OPTIONS = False
KINDS = ["synth", "noise", "flat", "extreme", "fastmotion", "subpel"]


def one_case(rng, lib, enc_tool, tmp, max_mbs, run=True, big=False):
    while True:
        w = int(rng.integers(8, 41)) * 2 if rng.random() < 0.5 else int(rng.integers(1, 26)) * 16
        h = int(rng.integers(8, 41)) * 2 if rng.random() < 0.5 else int(rng.integers(1, 20)) * 16
        if big:                                            # up to 1920x1088 / 4096x2304-class pictures
            w, h = int(rng.integers(40, 961)) * 2, int(rng.integers(30, 545)) * 2
        if w < 16 or h < 16:
            continue
        mbs = ((w + 15) // 16) * ((h + 15) // 16)
        if mbs <= max_mbs:
            break
    mb_h = (h + 15) // 16
    frames = int(rng.integers(2, 7))
    scene = int(rng.random() < 0.5)
    cut = -1
    if scene and mbs <= 150 and rng.random() < 0.3:        # long enough for the scene-change detector to be allowed to act
        frames = int(rng.integers(18, 30))
        cut = int(rng.integers(10, frames)) if rng.random() < 0.7 else -1
    aq, fskip = int(rng.random() < 0.3), int(rng.random() < 0.3)   # accepted and without effect, as in the reference
    # frame rate and bitrate only feed the level selection here; drawn from a side stream so that the picture / slice /
    # content draws of a (seed, index) pair stay what they were when the GPU tier was last run on them
    rng2 = np.random.default_rng([w, h, frames, mbs])
    fps = float(rng2.choice([30, 30, 5, 12.5, 15, 25, 60]))
    bitrate = int(rng2.choice([5000000, 5000000, 64000, 300000, 1200000, 20000000, 80000000]))
    if LEVEL1 and mbs <= 99:                               # --level1: level 1 / 1b streams (search range 63 at level 1)
        fps = float(rng2.choice([f for f in (1, 5, 7.5, 12.5, 15) if mbs * f <= 1485]))
        bitrate = int(rng2.choice([64000, 76000, 100000, 150000]))
    qp = int(rng.choice([0, 1, 5, 10, 12, 18, 24, 26, 30, 36, 40, 45, 51, int(rng.integers(0, 52))]))
    iper = int(rng.choice([0, 0, 0, 1, 2, 3]))
    cplx = int(rng.choice([0, 0, 1, 2]))
    idc = int(rng.choice([0, 0, 1, 2]))
    nsl = int(rng.integers(1, min(4, mb_h) + 1))
    kind = str(rng.choice(KINDS))
    odd = rng2_odd(w, h)
    if odd:                                                # odd picture sizes (the input layout is the console's: planes of
        w, h = w - odd[0], h - odd[1]                      # w*h and (w>>1)*(h>>1) bytes inside frames of w*h*3/2 bytes)
        kind = "oddsize"
    alpha, beta = (int(x) for x in rng.choice([0, 0, 0, -6, -3, 2, 6], 2))
    crop = int(rng.random() < 0.85)
    spsid = int(rng.random() < 0.7)
    if spsid == 0 and (w + h) % 3 == 0:                    # SPS_LISTING / SPS_LISTING_AND_PPS_INCREASING behave like CONSTANT_ID here
        spsid = 2 + (w // 2) % 2
    fidr = int(rng.integers(1, frames)) if rng.random() < 0.2 else -1
    # slice threads of the reference (stream effect: idc 0 -> 2); side stream again, see above
    threads = int(rng2.choice([1, 1, 1, 2, 4]))
    # (the threaded reference writes every slice into a buffer of its own: near the raw picture size its overflow
    #  behaviour differs from the single-threaded one that is modelled here, so those cases stay single-threaded)
    if threads > 1 and qp < (32 if (kind in ("noise", "extreme", "fastmotion") or cut > 0) else 16):
        threads = 1
    yuv = content(kind, w, h, frames, rng)
    if cut > 0:                                            # abrupt change of content at frame `cut`
        other = content(str(rng.choice(KINDS)), w, h, frames, rng)
        fsz = w * h * 3 // 2
        yuv = yuv[:cut * fsz] + other[cut * fsz:]
    params = dict(fMaxFrameRate=fps, iTargetBitrate=bitrate, iDLayerQp=qp, uiIntraPeriod=iper, iComplexityMode=cplx,
                  iLoopFilterDisableIdc=idc, iLoopFilterAlphaC0Offset=alpha, iLoopFilterBetaOffset=beta,
                  bEnableFrameCroppingFlag=crop, eSpsPpsIdStrategy=spsid, bEnableSceneChangeDetect=scene,
                  bEnableAdaptiveQuant=aq, bEnableFrameSkip=fskip, iMultipleThreadIdc=threads)
    flags = ["-rc", "-1", "-qp", str(qp), "-fps", str(fps), "-bitrate", str(bitrate), "-iper", str(iper), "-complexity", str(cplx), "-deblock", str(idc),
             "-alpha", str(alpha), "-beta", str(beta), "-crop", str(crop), "-spsid", str(spsid), "-forceidr", str(fidr),
             "-scene", str(scene), "-aq", str(aq), "-frameskip", str(fskip), "-threads", str(threads), "-loadbalancing", "0", "-quiet"]
    raster = -1
    if rng.random() < 0.2:                                 # SM_RASTER_SLICE: N macroblocks per slice, 0 = one slice per row
        raster = int(rng.choice([0, 0, int(rng.integers(1, mbs + 8)), int(rng.integers(max(1, mbs // 36), max(2, mbs // 2) + 1))]))
        if raster == 0 and mb_h > 35 or raster > 0 and (mbs + raster - 1) // raster > 35 + 1:
            raster = -1
    if raster >= 0:
        arr = (C.c_uint32 * 35)(*([raster] * 35))
        params.update(uiSliceMode=2, uiSliceMbNum=arr)
        flags += ["-slcmd", "2", "-slcmbnum", str(raster)]
        nsl = 100 + raster
    elif nsl > 1:
        params.update(uiSliceMode=1, uiSliceNum=nsl)
        flags += ["-slcmd", "1", "-slcnum", str(nsl)]
    options, psets = [], -1
    if OPTIONS and frames > 2:                             # --options: SetOption / EncodeParameterSets calls in mid-stream
        rng3 = np.random.default_rng([w, h, frames, qp, 77])
        if rng3.random() < 0.5:
            at, v = int(rng3.integers(1, frames)), int(rng3.choice([-1, 0, 1, 2, 3, 5]))
            options.append((at, oh.OPTION_IDR_INTERVAL, v)); flags += ["-setidr", str(at), str(v)]
        if rng3.random() < 0.5:
            at, v = int(rng3.integers(1, frames)), int(rng3.integers(0, 3))
            options.append((at, oh.OPTION_COMPLEXITY, v)); flags += ["-setcplx", str(at), str(v)]
        if rng3.random() < 0.3:
            at, v = int(rng3.integers(1, frames)), float(rng3.choice([1, 5, 7.5, 15, 30, 60, 100]))
            options.append((at, oh.OPTION_FRAME_RATE, v)); flags += ["-setfps", str(at), str(v)]
        if rng3.random() < 0.3:
            psets = int(rng3.integers(0, frames))
            flags += ["-paramsets", str(psets)]
    desc = "%dx%d f%d@%g/%dk t%d qp%d iper%d c%d idc%d a%d b%d crop%d id%d fi%d sc%d/%d sl%d %s" % (w, h, frames, fps, bitrate // 1000, threads, qp, iper, cplx, idc, alpha, beta, crop, spsid, fidr, scene, cut, nsl, kind)
    if options or psets >= 0:
        desc += " opt%s ps%d" % ([(a, o, v) for a, o, v in options], psets)
    if not run:                                            # --only: just keep the random stream in step
        return desc, "ok"
    fi, fo = os.path.join(tmp, "in.yuv"), os.path.join(tmp, "ref.264")
    open(fi, "wb").write(yuv)
    r = subprocess.run([enc_tool, "-i", fi, "-w", str(w), "-h", str(h), "-o", fo] + flags, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    ref_failed = r.returncode != 0
    ref = b"" if ref_failed else open(fo, "rb").read()
    try:
        bs, _ = oh.encode_sequence(yuv, w, h, lib_path=lib, force_idr_at=fidr, options_at=options, param_sets_at=psets, **params)
    except oh.WelsHipError as e:
        # the reference gives up with cmMallocMemeError (3) when a frame overflows its bitstream buffer even at QP 50
        if ref_failed and e.code == 3 and "EncodeFrame failed: 3" in r.stderr.decode():
            return desc, "ok (both refuse the frame: bitstream buffer overflow)"
        if ref_failed and e.code in (1, 4) and "Initialize failed" in r.stderr.decode():
            return desc, "ok (both reject the parameters)"
        return desc, "ERROR %s" % e
    if ref_failed:
        return desc, "MISMATCH reference failed (%s), ours encoded %d B" % (r.stderr.decode().strip().splitlines()[-1], len(bs))
    open(os.path.join(tmp, "ours.264"), "wb").write(bs)
    if bs == ref:
        return desc, "ok"
    n = min(len(bs), len(ref))
    first = next((i for i in range(n) if bs[i] != ref[i]), n)
    return desc, "MISMATCH ours %d B ref %d B first diff at %d" % (len(bs), len(ref), first)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-mbs", type=int, default=400)
    ap.add_argument("--hip", action="store_true")
    ap.add_argument("--emu-define", action="append", default=[], help="extra -D for the emulation build (candidate code paths)")
    ap.add_argument("--big", action="store_true", help="picture sizes up to 1920x1088 (use with --max-mbs 8200)")
    ap.add_argument("--level1", action="store_true", help="pictures of at most 99 MBs get a frame rate / bitrate that selects level 1 or 1b")
    ap.add_argument("--options", action="store_true", help="add SetOption (IDR interval, complexity) and EncodeParameterSets calls in mid-stream")
    ap.add_argument("--lib", default=None, help="use this shared library (e.g. a candidate build from build_hip(defines=...))")
    ap.add_argument("--only", type=int, default=-1, help="run just this case index of the seed (same random stream)")
    ap.add_argument("--keep", default=None, help="directory that keeps in.yuv / ref.264 / ours.264 of the last case run")
    a = ap.parse_args()
    global LEVEL1, OPTIONS
    LEVEL1, OPTIONS = a.level1, a.options
    enc_tool = os.path.join(ROOT, "oracle", "_ref", "ref_enc")
    if not os.path.exists(enc_tool):
        sys.exit("oracle/_ref/ref_enc not built (python -c 'import __graft_entry__ as g; g.build()')")
    lib = a.lib if a.lib else B.build_hip() if a.hip else B.build_emu(defines=tuple(a.emu_define), tag="_".join(d.lower() for d in a.emu_define))
    rng = np.random.default_rng(a.seed)
    bad = 0
    with tempfile.TemporaryDirectory() as tmpdir:
        tmp = a.keep or tmpdir
        os.makedirs(tmp, exist_ok=True)
        for i in range(a.cases if a.only < 0 else a.only + 1):
            desc, res = one_case(rng, lib, enc_tool, tmp, a.max_mbs, run=(a.only < 0 or i == a.only), big=a.big)
            if a.only >= 0 and i != a.only:
                continue
            if not res.startswith("ok"):
                bad += 1
            print("%4d %-92s %s" % (i, desc, res), flush=True)
    print("%d cases, %d failed" % (a.cases if a.only < 0 else 1, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
