"""Randomised parity of screen-content sessions through the dispatch-table binding (TEST INFRASTRUCTURE).

Synthetic "document" clips -- text-like blocks on a light background with a few noisy pictures, shown through a window that
stands still, scrolls vertically or horizontally by even offsets, or jumps, plus small local changes (a cursor) -- at random
sizes (also sizes that are not a multiple of 16), encoded by the unmodified reference (oracle/_ref/ref_enc) and by the
reference with this engine behind SWelsFuncPtrList (oracle/_ref/ref_enc_hip), `-usage 1` with random rate-control mode, slice
mode, complexity, temporal layers, LTR, denoising, deblocking mode and entropy coder.  The two bitstreams must be identical.
This is what found that HIGH complexity switches the static-skip decision (and with it the static-block map) off.

usage: fuzz_screen.py [--lib path] [--seed S] [--cases N] [--workers W] [--usage 0|1] [-v]
"""
import argparse
import os
import random
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def make_clip(w, h, n, seed):
    """n I420 frames of w x h: a window onto a tall synthetic document."""
    rng = np.random.default_rng(seed)
    H, W = h * 3, w + 64
    doc = np.full((H, W), 235, np.uint8)
    for _ in range(int(H * W / 900)):
        y = rng.integers(0, H - 12); x = rng.integers(0, W - 40)
        doc[y:y + rng.integers(1, 10), x:x + rng.integers(3, 40)] = rng.integers(0, 90)
    for _ in range(6):
        y = rng.integers(0, H - 64); x = rng.integers(0, W - 64)
        doc[y:y + 64, x:x + 64] = rng.integers(0, 255, (64, 64))
    cu = np.full((H // 2, W // 2), 128, np.uint8)
    cv = np.full((H // 2, W // 2), 128, np.uint8)
    for _ in range(20):
        y = rng.integers(0, H // 2 - 20); x = rng.integers(0, W // 2 - 20)
        cu[y:y + 20, x:x + 20] = rng.integers(60, 200); cv[y:y + 20, x:x + 20] = rng.integers(60, 200)
    frames = []
    y0 = int(rng.integers(0, h)) & ~1
    x0 = 16
    for _ in range(n):
        mode = rng.integers(0, 6)
        if mode in (2, 3):
            y0 = int(np.clip(y0 + 2 * rng.integers(-20, 40), 0, H - h - 2))      # vertical scroll
        elif mode == 4:
            x0 = int(np.clip(x0 + 2 * rng.integers(-8, 8), 0, W - w - 2))        # horizontal scroll
        elif mode == 5:
            y0 = int(rng.integers(0, H - h - 2)) & ~1                            # jump
        Y = doc[y0:y0 + h, x0:x0 + w].copy()
        U = cu[y0 // 2:y0 // 2 + h // 2, x0 // 2:x0 // 2 + w // 2].copy()
        V = cv[y0 // 2:y0 // 2 + h // 2, x0 // 2:x0 // 2 + w // 2].copy()
        if rng.integers(0, 3) == 0:
            yy = rng.integers(0, h - 8); xx = rng.integers(0, w - 8)
            Y[yy:yy + 8, xx:xx + 8] = rng.integers(0, 255, (8, 8))
        frames.append(Y.tobytes() + U.tobytes() + V.tobytes())
    return b"".join(frames)


QP_MAX = None       # --qp-max N: constant-QP cases only, QP 0..N (the CAVLC overflow / TRY_REENCODING regime of noisy content)


def case_flags(seed, usage=1):
    rnd = random.Random(seed)
    w = rnd.choice([64, 96, 160, 176, 320, 322, 400, 640]) & ~1
    h = rnd.choice([48, 64, 96, 144, 146, 240, 360]) & ~1
    n = rnd.randint(6, 14)
    flags = ["-usage", str(usage), "-fps", str(rnd.choice([10, 15, 30]))]
    rc = rnd.choice([-1, -1, 1, 0, 3])
    qp = rnd.randint(10, 40)
    if QP_MAX is not None:
        rc, qp = -1, rnd.randint(0, QP_MAX)
    flags += ["-rc", str(rc)]
    flags += ["-qp", str(qp)] if rc == -1 else ["-bitrate", str(rnd.choice([100000, 400000, 1500000]))]
    sl = rnd.choice([0, 1, 1, 2])
    flags += ["-slcmd", str(sl)]
    if sl == 1:
        flags += ["-slcnum", str(rnd.choice([2, 3, 4]))]
    if sl == 2:
        flags += ["-slcmbnum", str(rnd.choice([7, 13, 40]))]
    flags += ["-complexity", str(rnd.choice([0, 0, 1, 2])), "-ltr", str(rnd.choice([0, 1])), "-numtl", str(rnd.choice([1, 1, 2, 3])), "-scene", "1",
              "-denoise", str(rnd.choice([0, 1])), "-deblock", str(rnd.choice([0, 0, 1, 2]))]
    if rnd.random() < 0.3:
        flags += ["-cabac", "1", "-profile", "77"]
    if usage == 0:
        flags += ["-bgd", str(rnd.choice([0, 1]))]
    return w, h, n, flags


def run_case(seed, lib, workdir, usage=1):
    """-> (seed, verdict, detail): verdict "ok", "invalid" (the reference rejects the parameters) or "DIFF"."""
    w, h, n, flags = case_flags(seed, usage)
    f = os.path.join(workdir, "c%d.yuv" % seed)
    a, b = os.path.join(workdir, "a%d.264" % seed), os.path.join(workdir, "b%d.264" % seed)
    open(f, "wb").write(make_clip(w, h, n, seed))
    base = ["-i", f, "-w", str(w), "-h", str(h), "-quiet"] + flags
    try:
        p = subprocess.run([os.path.join(REF, "ref_enc"), "-o", a] + base, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if p.returncode:
            return seed, "invalid", flags
        env = dict(os.environ, WELSHIP_LIB=lib, WELS_HIP_GOM="1", WELS_HIP_CHECK_BITS="1", WELS_HIP_TRACE="1")
        q = subprocess.run([os.path.join(REF, "ref_enc_hip"), "-o", b] + base, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        err = q.stderr.decode(errors="replace")
        pics = err.count("welship hooks: did")
        same = q.returncode == 0 and open(a, "rb").read() == open(b, "rb").read() and "welship hooks: installed" in err and pics >= 1
        return seed, "ok" if same else "DIFF", (w, h, n, pics, flags, [l for l in err.splitlines() if "welship" in l and "did" not in l][-2:])
    finally:
        for x in (f, a, b):
            if os.path.exists(x):
                os.remove(x)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "openh264_amd", "libwelship.so"))
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=32)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--usage", type=int, default=1, help="1 screen content (default), 0 camera video on the same clips (scene-change I pictures in mid-stream)")
    ap.add_argument("--qp-max", type=int, default=None, help="constant-QP cases only, QP 0..N")
    ap.add_argument("-v", action="store_true")
    a = ap.parse_args()
    global QP_MAX
    QP_MAX = a.qp_max
    a.lib = os.path.abspath(a.lib)
    bad = ok = invalid = 0
    with tempfile.TemporaryDirectory() as d, ThreadPoolExecutor(a.workers) as ex:
        for r in ex.map(lambda s: run_case(s, a.lib, d, a.usage), range(a.seed, a.seed + a.cases)):
            if r[1] == "DIFF":
                bad += 1
                print(r)
            else:
                ok += r[1] == "ok"
                invalid += r[1] == "invalid"
                if a.v:
                    print(r)
    print(("screen-content" if a.usage == 1 else "camera-video") + " fuzz: seeds %d..%d, identical %d, rejected by the reference %d, different %d, library %s" % (a.seed, a.seed + a.cases - 1, ok, invalid, bad, os.path.basename(a.lib)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
