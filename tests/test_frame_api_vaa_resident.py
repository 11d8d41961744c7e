"""WelsHipFrameVaa leaves the picture it analysed on the device, and the WelsHipFrameEncode that follows skips the upload -- but only while
the caller's buffer still holds what was uploaded: a buffer's address does not identify its content (a caller may drop the analysed picture,
refill the buffer and encode that).  The records of an encode after such a refill must be those of the NEW content."""
import ctypes as C
import random

import pytest

import openh264_amd as oh
from test_frame_api_retry import FrameCfg, FrameJob, REC_BYTES


class VaaJob(C.Structure):            # WelsHipVaaJob, field for field
    _fields_ = [("pCur", C.c_void_p * 3), ("iCurStride", C.c_int32 * 3), ("pRef", C.c_void_p * 3), ("iRefStride", C.c_int32 * 3),
                ("iPicWidth", C.c_int32), ("iPicHeight", C.c_int32), ("bCalcVar", C.c_int32), ("bCalcBgd", C.c_int32), ("bCalcSsd", C.c_int32),
                ("pSad8x8", C.POINTER(C.c_int32)), ("pSsd16x16", C.POINTER(C.c_int32)), ("pSum16x16", C.POINTER(C.c_int32)),
                ("pSumOfSquare16x16", C.POINTER(C.c_int32)), ("pSumOfDiff8x8", C.POINTER(C.c_int32)), ("pMad8x8", C.POINTER(C.c_uint8)),
                ("pFrameSad", C.POINTER(C.c_int32))]


def _encode_after_refill(lib_path):
    lib = oh.load_library(lib_path)
    lib.WelsHipFrameCtxCreate.argtypes = [C.POINTER(C.c_void_p), C.POINTER(FrameCfg)]
    lib.WelsHipFrameCtxDestroy.argtypes = [C.c_void_p]
    lib.WelsHipFrameCtxDestroy.restype = None
    lib.WelsHipFrameEncode.argtypes = [C.c_void_p, C.POINTER(FrameJob), C.POINTER(C.c_void_p)]
    lib.WelsHipFrameVaa.argtypes = [C.c_void_p, C.POINTER(VaaJob)]
    w, h = 96, 64
    mbs = (w // 16) * (h // 16)
    rnd = random.Random(3)

    def picture(seed):
        r = random.Random(seed)
        f = bytearray(w * h * 3 // 2)
        for i in range(w * h):
            f[i] = (80 + 50 * (((i % w) // 8 + (i // w) // 8) & 1) + r.randint(0, 20)) & 255
        for i in range(w * h, len(f)):
            f[i] = 128
        return f

    def planes(buf):
        base = C.addressof(buf)
        return base, base + w * h, base + w * h + (w // 2) * (h // 2)

    first = (C.c_int32 * 2)(0, mbs)

    def encode(ctx, buf, cur, ref):
        j = FrameJob()
        j.cbSize = C.sizeof(FrameJob)
        j.iCurPic, j.iRefPic, j.eSliceType, j.iQp, j.iComplexityMode, j.iMvRange = cur, ref, (2 if ref < 0 else 0), 26, 1, 64
        j.iNumSlices, j.pSliceFirstMb = 1, first
        j.iDeblockIdc, j.bDeblock, j.bExpand = 0, 1, 1
        j.pSrc[0], j.pSrc[1], j.pSrc[2] = planes(buf)
        j.iSrcStride[0], j.iSrcStride[1], j.iSrcStride[2] = w, w // 2, w // 2
        j.iNumRefIdxL0Active = 1
        rec = C.c_void_p()
        rc = lib.WelsHipFrameEncode(ctx, C.byref(j), C.byref(rec))
        assert rc == 0, (rc, lib.WelsHipGetLastError())
        return C.string_at(rec, REC_BYTES * mbs)

    def vaa(ctx, cur, ref):
        v = VaaJob()
        v.pCur[0], v.pCur[1], v.pCur[2] = planes(cur)
        v.pRef[0], v.pRef[1], v.pRef[2] = planes(ref)
        for i, s in enumerate((w, w // 2, w // 2)):
            v.iCurStride[i] = s
            v.iRefStride[i] = s
        v.iPicWidth, v.iPicHeight = w, h
        sad, fs = (C.c_int32 * (4 * mbs))(), C.c_int32()
        v.pSad8x8, v.pFrameSad = sad, C.pointer(fs)
        assert lib.WelsHipFrameVaa(ctx, C.byref(v)) == 0, lib.WelsHipGetLastError()
        return fs.value

    p0, p1, p2 = picture(1), picture(2), picture(5)
    results = []
    for refill in (False, True):
        ctx = C.c_void_p()
        cfg = FrameCfg(0, w, h, 3)
        assert lib.WelsHipFrameCtxCreate(C.byref(ctx), C.byref(cfg)) == 0, lib.WelsHipGetLastError()
        a = (C.c_uint8 * len(p0)).from_buffer_copy(bytes(p0))
        b = (C.c_uint8 * len(p0)).from_buffer_copy(bytes(p1 if refill else p2))
        encode(ctx, a, 0, -1)
        assert vaa(ctx, b, a) > 0                 # the analysed picture is resident now ...
        if refill:
            C.memmove(b, bytes(p2), len(p2))      # ... the caller drops it and refills the buffer
        results.append(encode(ctx, b, 1, 0))
        lib.WelsHipFrameCtxDestroy(ctx)
    assert results[0] == results[1]               # both contexts encoded p2 against p0


def test_encode_after_refill_of_the_analysed_buffer(emu_lib):
    _encode_after_refill(emu_lib)


@pytest.mark.gpu
def test_hip_encode_after_refill_of_the_analysed_buffer(hip_lib):
    _encode_after_refill(hip_lib)
