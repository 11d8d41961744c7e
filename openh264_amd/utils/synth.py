"""Deterministic synthetic I420 test sequences (SURVEY.md 8(d), config 3 generator).

frame 0 = LCG noise low-passed with an 8x8 box filter (seed 0x1234); frame n = frame 0 translated
by (3n, 2n) pixels (wrap-around) plus fresh per-frame LCG noise in [-4, 4]; chroma = 128 +- 16
gradients that drift with n.  Pure numpy integer arithmetic, so the bytes are identical on every
machine -- the golden SHA1s under tests/golden/ were produced from exactly these bytes.
"""
import numpy as np


def _lcg(n, seed):
    # 32-bit LCG (Numerical Recipes constants), vectorised through its closed form per block
    out = np.empty(n, dtype=np.uint32)
    a, c = np.uint64(1664525), np.uint64(1013904223)
    x = np.uint64(seed & 0xFFFFFFFF)
    # generate in chunks with a python loop over 4096-element strides using the jump-ahead recurrence
    blk = 4096
    # precompute A^k, C_k for k = 1..blk
    ak = np.empty(blk, dtype=np.uint64)
    ck = np.empty(blk, dtype=np.uint64)
    aa, cc = np.uint64(1), np.uint64(0)
    mask = np.uint64(0xFFFFFFFF)
    for i in range(blk):
        aa = (aa * a) & mask
        cc = (cc * a + c) & mask
        ak[i], ck[i] = aa, cc
    pos = 0
    while pos < n:
        m = min(blk, n - pos)
        vals = (ak[:m] * x + ck[:m]) & mask
        out[pos:pos + m] = vals.astype(np.uint32)
        x = vals[m - 1]
        pos += m
    return out


def synth_sequence(width, height, frames, seed=0x1234):
    """Returns bytes of `frames` I420 frames of width x height."""
    w, h = width, height
    base = (_lcg(w * h, seed) >> 24).astype(np.int32).reshape(h, w)
    # 8x8 box low-pass with wrap-around, integer
    acc = np.zeros_like(base)
    for dy in range(8):
        for dx in range(8):
            acc += np.roll(np.roll(base, dy, axis=0), dx, axis=1)
    f0 = acc >> 6
    # add some structure: blocks + a ramp so that intra modes / ME have something to find
    yy, xx = np.mgrid[0:h, 0:w]
    f0 = (f0 * 3 // 4 + ((xx // 24 + yy // 16) % 5) * 12 + (xx * 40 // max(w, 1))).astype(np.int32)
    f0 = np.clip(f0, 0, 255)
    out = bytearray()
    for n in range(frames):
        y = np.roll(np.roll(f0, 2 * n, axis=0), 3 * n, axis=1)
        noise = ((_lcg(w * h, seed + 7919 * (n + 1)) >> 16) % 9).astype(np.int32).reshape(h, w) - 4
        y = np.clip(y + noise, 0, 255).astype(np.uint8)
        cy, cx = np.mgrid[0:h // 2, 0:w // 2]
        u = (128 + ((cx + 2 * n) * 32 // max(w // 2, 1)) - 16).astype(np.uint8)
        v = (128 + ((cy + n) * 32 // max(h // 2, 1)) - 16).astype(np.uint8)
        out += y.tobytes() + u.tobytes() + v.tobytes()
    return bytes(out)


def checker_sequence(width, height, frames, period=3):
    """Saturated 0/255 checkerboards that drift by (3, 1) pixels per frame, chroma likewise: residuals with the
    largest levels the transform can produce (CAVLC level-escape overflow at very low QP, clipping paths)."""
    w, h = width, height
    yy, xx = np.mgrid[0:h, 0:w]
    out = bytearray()
    for n in range(frames):
        y = ((((xx + n * 3) // period + (yy + n) // period) & 1) * 255).astype(np.uint8)
        c = ((((xx[:h // 2, :w // 2] + n) // period) & 1) * 255).astype(np.uint8)
        out += y.tobytes() + c.tobytes() + (255 - c).tobytes()
    return bytes(out)


def pan_sequence(width, height, frames, dx):
    """A smooth texture moving `dx` samples per picture to the right (wrapping): the true motion sits at the edge of
    the integer search range, which level 1 narrows to 63 samples."""
    rng = np.random.default_rng(width * 131 + height * 7 + dx)
    base = rng.integers(0, 256, (height, width)).astype(np.int32)
    acc = np.zeros_like(base)
    for d in range(4):
        acc += np.roll(base, d, axis=1) + np.roll(base, d, axis=0)
    tex = (acc >> 3).astype(np.uint8)
    out = bytearray()
    for n in range(frames):
        out += np.roll(tex, dx * n, axis=1).tobytes()
        cu = np.roll(tex[::2, ::2], dx * n // 2, axis=1)
        out += cu.tobytes()
        out += (255 - cu).tobytes()
    return bytes(out)


def make_sequence(content, width, height, frames):
    """`content`: "synth" (default generator), "checker<period>" or "pan<dx>"."""
    if content.startswith("checker"):
        return checker_sequence(width, height, frames, int(content[7:] or 3))
    if content.startswith("pan"):
        return pan_sequence(width, height, frames, int(content[3:] or 64))
    return synth_sequence(width, height, frames)
