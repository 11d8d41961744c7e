"""openh264_amd -- Python binding (ctypes) over libwelship.so, the MI355X-native macroblock engine
behind the OpenH264 encoder API.

`Encoder` mirrors the reference's ISVCEncoder (codec/api/wels/codec_api.h:272-343): same method
names, argument meaning and error behaviour (0 = cmResultSuccess, non-zero = CM_RETURN error), so
the parity tests read like test/api/BaseEncoderTest.cpp.  The product library is HIP-only: loading
works anywhere libamdhip64 is installed, but any call that needs the device fails loudly when no
MI355X is present -- there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libwelship.so")

videoFormatI420 = 23
RC_OFF_MODE = -1
(LOW_COMPLEXITY, MEDIUM_COMPLEXITY, HIGH_COMPLEXITY) = (0, 1, 2)
(SM_SINGLE_SLICE, SM_FIXEDSLCNUM_SLICE) = (0, 1)
(CONSTANT_ID, INCREASING_ID) = (0, 1)
(videoFrameTypeInvalid, videoFrameTypeIDR, videoFrameTypeI, videoFrameTypeP, videoFrameTypeSkip) = range(5)
cmResultSuccess, cmInitParaError, cmUnknownReason, cmUnsupportedData = 0, 1, 2, 4
ERR_NO_DEVICE, ERR_VLC_OVERFLOW = 100, 101
HAS_INTER_PATH = True


OPTION_DATAFORMAT, OPTION_IDR_INTERVAL, OPTION_FRAME_RATE, OPTION_COMPLEXITY = 0, 1, 4, 15     # ENCODER_OPTION ids


class SEncParamExt(C.Structure):
    """WelsHipEncParam (include/welship.h) -- the honoured subset of SEncParamExt, same field names."""
    _fields_ = [
        ("iUsageType", C.c_int32), ("iPicWidth", C.c_int32), ("iPicHeight", C.c_int32),
        ("iTargetBitrate", C.c_int32), ("iRCMode", C.c_int32), ("fMaxFrameRate", C.c_float),
        ("iTemporalLayerNum", C.c_int32), ("iSpatialLayerNum", C.c_int32), ("iComplexityMode", C.c_int32),
        ("uiIntraPeriod", C.c_uint32), ("eSpsPpsIdStrategy", C.c_int32), ("iEntropyCodingModeFlag", C.c_int32),
        ("iLoopFilterDisableIdc", C.c_int32), ("iLoopFilterAlphaC0Offset", C.c_int32), ("iLoopFilterBetaOffset", C.c_int32),
        ("bEnableFrameCroppingFlag", C.c_int32), ("iDLayerQp", C.c_int32), ("uiSliceMode", C.c_int32), ("uiSliceNum", C.c_int32),
        ("bEnableAdaptiveQuant", C.c_int32), ("bEnableBackgroundDetection", C.c_int32), ("bEnableSceneChangeDetect", C.c_int32),
        ("bEnableLongTermReference", C.c_int32), ("bEnableDenoise", C.c_int32), ("bEnableFrameSkip", C.c_int32),
        ("iDevice", C.c_int32), ("iMultipleThreadIdc", C.c_int32), ("reserved", C.c_int32 * 6), ("uiSliceMbNum", C.c_uint32 * 35),
    ]


class SSourcePicture(C.Structure):
    _fields_ = [("iColorFormat", C.c_int32), ("iStride", C.c_int32 * 4), ("pData", C.c_void_p * 4),
                ("iPicWidth", C.c_int32), ("iPicHeight", C.c_int32), ("uiTimeStamp", C.c_int64)]


class SLayerBSInfo(C.Structure):
    _fields_ = [("uiTemporalId", C.c_uint8), ("uiSpatialId", C.c_uint8), ("uiQualityId", C.c_uint8),
                ("eFrameType", C.c_int32), ("uiLayerType", C.c_uint8), ("iSubSeqId", C.c_int32),
                ("iNalCount", C.c_int32), ("pNalLengthInByte", C.POINTER(C.c_int32)), ("pBsBuf", C.POINTER(C.c_uint8))]


class SFrameBSInfo(C.Structure):
    _fields_ = [("iLayerNum", C.c_int32), ("sLayerInfo", SLayerBSInfo * 4), ("eFrameType", C.c_int32),
                ("iFrameSizeInBytes", C.c_int32), ("uiTimeStamp", C.c_int64)]


class WelsHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("welship error %d: %s" % (code, msg))
        self.code = code


_libs = {}


def load_library(path=None):
    """Load libwelship.so (or, in the CPU-only test tier, the emulation test build given by `path`)."""
    path = path or os.environ.get("WELSHIP_LIB") or DEFAULT_LIB
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError("%s not found -- run `python -c 'import __graft_entry__ as g; g.build()'` first" % path)
    lib = C.CDLL(path)
    lib.WelsHipCreateEncoder.argtypes = [C.POINTER(C.c_void_p)]
    lib.WelsHipDestroyEncoder.argtypes = [C.c_void_p]
    lib.WelsHipDestroyEncoder.restype = None
    lib.WelsHipGetDefaultParams.argtypes = [C.c_void_p, C.POINTER(SEncParamExt)]
    lib.WelsHipInitializeExt.argtypes = [C.c_void_p, C.POINTER(SEncParamExt)]
    lib.WelsHipUninitialize.argtypes = [C.c_void_p]
    lib.WelsHipEncodeFrame.argtypes = [C.c_void_p, C.POINTER(SSourcePicture), C.POINTER(SFrameBSInfo)]
    lib.WelsHipForceIntraFrame.argtypes = [C.c_void_p, C.c_int]
    lib.WelsHipGetReconFrame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.WelsHipBackendName.argtypes = [C.c_void_p]
    lib.WelsHipBackendName.restype = C.c_char_p
    lib.WelsHipGetLastError.restype = C.c_char_p
    _libs[path] = lib
    return lib


class Encoder:
    """Mirror of ISVCEncoder for the hot path (create = WelsCreateSVCEncoder)."""

    def __init__(self, lib_path=None):
        self._lib = load_library(lib_path)
        h = C.c_void_p()
        rc = self._lib.WelsHipCreateEncoder(C.byref(h))
        if rc:
            raise WelsHipError(rc, "WelsHipCreateEncoder")
        self._h = h
        self._w = self._h_pix = 0

    def _err(self):
        return (self._lib.WelsHipGetLastError() or b"").decode()

    def GetDefaultParams(self):
        p = SEncParamExt()
        self._lib.WelsHipGetDefaultParams(self._h, C.byref(p))
        return p

    def InitializeExt(self, param):
        rc = self._lib.WelsHipInitializeExt(self._h, C.byref(param))
        if rc == 0:
            self._w, self._h_pix = param.iPicWidth, param.iPicHeight
        return rc

    def Uninitialize(self):
        return self._lib.WelsHipUninitialize(self._h)

    def SetOption(self, option_id, value):
        """ISVCEncoder::SetOption for the int / float options of include/welship.h (ENCODER_OPTION ids)."""
        v = C.c_float(value) if option_id == OPTION_FRAME_RATE else C.c_int32(int(value))
        self._lib.WelsHipSetOption.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        return self._lib.WelsHipSetOption(self._h, option_id, C.byref(v))

    def GetOption(self, option_id):
        v = C.c_float(0) if option_id == OPTION_FRAME_RATE else C.c_int32(0)
        self._lib.WelsHipGetOption.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        rc = self._lib.WelsHipGetOption(self._h, option_id, C.byref(v))
        return rc, v.value

    def EncodeParameterSets(self):
        """ISVCEncoder::EncodeParameterSets: returns (rc, bytes of SPS + PPS)."""
        info = SFrameBSInfo()
        self._lib.WelsHipEncodeParameterSets.argtypes = [C.c_void_p, C.POINTER(SFrameBSInfo)]
        rc = self._lib.WelsHipEncodeParameterSets(self._h, C.byref(info))
        if rc:
            return rc, b""
        L = info.sLayerInfo[0]
        return 0, C.string_at(L.pBsBuf, sum(L.pNalLengthInByte[k] for k in range(L.iNalCount)))

    def ForceIntraFrame(self, idr=True):
        return self._lib.WelsHipForceIntraFrame(self._h, 1 if idr else 0)

    def EncodeFrame(self, yuv, timestamp=0):
        """yuv: bytes-like I420 frame of iPicWidth x iPicHeight.  Returns (rc, frame_type, bytes, nal_lengths)."""
        w, h = self._w, self._h_pix
        buf = (C.c_uint8 * len(yuv)).from_buffer_copy(yuv) if not isinstance(yuv, C.Array) else yuv
        base = C.addressof(buf)
        pic = SSourcePicture()
        pic.iColorFormat = videoFormatI420
        pic.iStride[0], pic.iStride[1], pic.iStride[2] = w, w // 2, w // 2
        pic.pData[0], pic.pData[1], pic.pData[2] = base, base + w * h, base + w * h + (w // 2) * (h // 2)
        pic.iPicWidth, pic.iPicHeight, pic.uiTimeStamp = w, h, timestamp
        info = SFrameBSInfo()
        rc = self._lib.WelsHipEncodeFrame(self._h, C.byref(pic), C.byref(info))
        if rc:
            return rc, videoFrameTypeInvalid, b"", []
        out = bytearray()
        nals = []
        for li in range(info.iLayerNum):
            L = info.sLayerInfo[li]
            n = sum(L.pNalLengthInByte[k] for k in range(L.iNalCount))
            nals += [L.pNalLengthInByte[k] for k in range(L.iNalCount)]
            out += C.string_at(L.pBsBuf, n)
        return 0, info.eFrameType, bytes(out), nals

    def GetReconFrame(self):
        n = self._w * self._h_pix * 3 // 2
        buf = (C.c_uint8 * n)()
        rc = self._lib.WelsHipGetReconFrame(self._h, buf, n)
        if rc:
            raise WelsHipError(rc, self._err())
        return bytes(buf)

    def backend_name(self):
        return (self._lib.WelsHipBackendName(self._h) or b"").decode()

    def overflow_reencodes(self):
        """Pictures re-encoded after a CAVLC level overflow since InitializeExt (developer statistic)."""
        self._lib.WelsHipDebugGetOverflowReencodes.argtypes = [C.c_void_p]
        return self._lib.WelsHipDebugGetOverflowReencodes(self._h)

    def last_error(self):
        return self._err()

    def close(self):
        if self._h:
            self._lib.WelsHipDestroyEncoder(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def encode_sequence(yuv_bytes, width, height, lib_path=None, stats=None, force_idr_at=-1, options_at=(), param_sets_at=-1, **params):
    """Convenience: encode a whole I420 sequence; returns (bitstream bytes, last recon frame).
    `stats`: optional dict that receives developer statistics (overflow_reencodes);
    `force_idr_at`: ForceIntraFrame(true) is called before that frame index;
    `options_at`: (frame index, option id, value) triples -> SetOption before that frame."""
    enc = Encoder(lib_path)
    p = enc.GetDefaultParams()
    p.iPicWidth, p.iPicHeight = width, height
    for k, v in params.items():
        if isinstance(v, (list, tuple)):          # array fields (uiSliceMbNum)
            arr = getattr(p, k)
            for i, x in enumerate(v):
                arr[i] = x
        else:
            setattr(p, k, v)
    rc = enc.InitializeExt(p)
    if rc:
        raise WelsHipError(rc, enc.last_error())
    fsz = width * height * 3 // 2
    out = bytearray()
    for i in range(len(yuv_bytes) // fsz):
        if i == force_idr_at:
            enc.ForceIntraFrame(True)
        if i == param_sets_at:
            rc, ps = enc.EncodeParameterSets()
            if rc:
                raise WelsHipError(rc, "EncodeParameterSets")
            out += ps
        for f, oid, val in options_at:
            if f == i and enc.SetOption(oid, val):
                raise WelsHipError(1, "SetOption(%d)" % oid)
        rc, _, bs, _ = enc.EncodeFrame(yuv_bytes[i * fsz:(i + 1) * fsz], timestamp=i * 33)
        if rc:
            raise WelsHipError(rc, enc.last_error())
        out += bs
    recon = enc.GetReconFrame()
    if stats is not None:
        stats["overflow_reencodes"] = enc.overflow_reencodes()
    enc.Uninitialize()
    enc.close()
    return bytes(out), recon


class EncoderGroup:
    """N independent sessions advancing in lock step (WelsHipGroup* in include/welship.h)."""

    def __init__(self, param, sessions, ring_slots=1, host_threads=1, lib_path=None):
        self._lib = lib = load_library(lib_path)
        if not hasattr(lib, "_group_ready"):
            lib.WelsHipGroupCreate.argtypes = [C.POINTER(C.c_void_p), C.POINTER(SEncParamExt), C.c_int, C.c_int, C.c_int]
            lib.WelsHipGroupDestroy.argtypes = [C.c_void_p]
            lib.WelsHipGroupDestroy.restype = None
            lib.WelsHipGroupUploadSource.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(SSourcePicture)]
            lib.WelsHipGroupBegin.argtypes = [C.c_void_p, C.c_int]
            lib.WelsHipGroupRunDevice.argtypes = [C.c_void_p, C.c_int]
            lib.WelsHipGroupFinish.argtypes = [C.c_void_p, C.POINTER(SFrameBSInfo)]
            lib.WelsHipGroupStepDeviceOnly.argtypes = [C.c_void_p, C.c_int]
            lib.WelsHipGroupGetReconFrame.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
            lib.WelsHipGroupBackendName.argtypes = [C.c_void_p]
            lib.WelsHipGroupBackendName.restype = C.c_char_p
            lib.WelsHipGroupBench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
            lib.WelsHipGroupEncodeFrames.argtypes = [C.c_void_p, C.POINTER(SSourcePicture), C.POINTER(SFrameBSInfo)]
            lib.WelsHipGroupHostStats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
            lib.WelsHipGroupSetPipelined.argtypes = [C.c_void_p, C.c_int]
            lib.WelsHipGroupEncodeFramesPipelined.argtypes = [C.c_void_p, C.POINTER(SSourcePicture), C.POINTER(SFrameBSInfo), C.POINTER(C.c_int)]
            lib._group_ready = True
        h = C.c_void_p()
        rc = lib.WelsHipGroupCreate(C.byref(h), C.byref(param), sessions, ring_slots, host_threads)
        if rc:
            raise WelsHipError(rc, (lib.WelsHipGetLastError() or b"").decode())
        self._h, self.n, self.w, self.h = h, sessions, param.iPicWidth, param.iPicHeight
        self._keep = []

    def upload(self, session, slot, yuv):
        w, h = self.w, self.h
        buf = (C.c_uint8 * len(yuv)).from_buffer_copy(yuv)
        base = C.addressof(buf)
        pic = SSourcePicture()
        pic.iColorFormat = videoFormatI420
        pic.iStride[0], pic.iStride[1], pic.iStride[2] = w, w // 2, w // 2
        pic.pData[0], pic.pData[1], pic.pData[2] = base, base + w * h, base + w * h + (w // 2) * (h // 2)
        pic.iPicWidth, pic.iPicHeight = w, h
        rc = self._lib.WelsHipGroupUploadSource(self._h, session, slot, C.byref(pic))
        if rc:
            raise WelsHipError(rc, "upload")

    def step(self, slot=0):
        """One full frame step for all sessions; returns the list of per-session bitstreams."""
        lib = self._lib
        rc = lib.WelsHipGroupBegin(self._h, slot)
        if rc == 0:
            rc = lib.WelsHipGroupRunDevice(self._h, 0)
        infos = (SFrameBSInfo * self.n)()
        if rc == 0:
            rc = lib.WelsHipGroupFinish(self._h, infos)
        if rc:
            raise WelsHipError(rc, (lib.WelsHipGetLastError() or b"").decode())
        out = []
        for info in infos:
            b = bytearray()
            for li in range(info.iLayerNum):
                L = info.sLayerInfo[li]
                b += C.string_at(L.pBsBuf, sum(L.pNalLengthInByte[k] for k in range(L.iNalCount)))
            out.append(bytes(b))
        return out

    def make_pictures(self, yuvs):
        """Host-resident source pictures for encode_frames: one I420 frame (bytes) per session.  Returns an opaque object
        that keeps the buffers alive."""
        w, h = self.w, self.h
        pics = (SSourcePicture * self.n)()
        keep = []
        for i, yuv in enumerate(yuvs):
            buf = (C.c_uint8 * len(yuv)).from_buffer_copy(yuv)
            keep.append(buf)
            base = C.addressof(buf)
            pic = pics[i]
            pic.iColorFormat = videoFormatI420
            pic.iStride[0], pic.iStride[1], pic.iStride[2] = w, w // 2, w // 2
            pic.pData[0], pic.pData[1], pic.pData[2] = base, base + w * h, base + w * h + (w // 2) * (h // 2)
            pic.iPicWidth, pic.iPicHeight = w, h
        return (pics, keep)

    def encode_frames(self, pictures, want_bytes=False):
        """WelsHipGroupEncodeFrames: one complete EncodeFrame for every session -- source upload, device passes, D2H of the
        MB records, host entropy coding.  Returns the total number of bitstream bytes (or the per-session bitstreams)."""
        infos = (SFrameBSInfo * self.n)()
        rc = self._lib.WelsHipGroupEncodeFrames(self._h, pictures[0], infos)
        if rc:
            raise WelsHipError(rc, (self._lib.WelsHipGetLastError() or b"").decode())
        if not want_bytes:
            return sum(info.iFrameSizeInBytes for info in infos)
        out = []
        for info in infos:
            b = bytearray()
            for li in range(info.iLayerNum):
                L = info.sLayerInfo[li]
                b += C.string_at(L.pBsBuf, sum(L.pNalLengthInByte[k] for k in range(L.iNalCount)))
            out.append(bytes(b))
        return out

    def set_pipelined(self, steps_ahead=1):
        """WelsHipGroupSetPipelined: encode_frames_pipelined from now on (before the first picture); the device runs up to
        `steps_ahead` frame steps ahead of the entropy coder."""
        rc = self._lib.WelsHipGroupSetPipelined(self._h, steps_ahead)
        if rc:
            raise WelsHipError(rc, (self._lib.WelsHipGetLastError() or b"").decode())

    def encode_frames_pipelined(self, pictures, want_bytes=False):
        """WelsHipGroupEncodeFramesPipelined: submits `pictures` (None: nothing, only finish) and finishes the oldest pending step once
        `steps_ahead` are pending.  Returns None when no step was finished, else what encode_frames returns -- for that EARLIER step."""
        infos = (SFrameBSInfo * self.n)()
        done = C.c_int(0)
        rc = self._lib.WelsHipGroupEncodeFramesPipelined(self._h, pictures[0] if pictures is not None else None, infos, C.byref(done))
        if rc:
            raise WelsHipError(rc, (self._lib.WelsHipGetLastError() or b"").decode())
        if not done.value:
            return None
        if not want_bytes:
            return sum(info.iFrameSizeInBytes for info in infos)
        out = []
        for info in infos:
            b = bytearray()
            for li in range(info.iLayerNum):
                L = info.sLayerInfo[li]
                b += C.string_at(L.pBsBuf, sum(L.pNalLengthInByte[k] for k in range(L.iNalCount)))
            out.append(bytes(b))
        return out

    def bench(self, steps, warmup):
        """Device-only hot path: returns dict(total_ms, md_ms, deblock_ms, expand_ms) from HIP events."""
        out = (C.c_double * 4)()
        rc = self._lib.WelsHipGroupBench(self._h, steps, warmup, out)
        if rc:
            raise WelsHipError(rc, (self._lib.WelsHipGetLastError() or b"").decode())
        return dict(total_ms=out[0], md_ms=out[1], deblock_ms=out[2], expand_ms=out[3])

    def host_stats(self):
        """Host share of the complete frame steps so far (thread time per picture)."""
        out = (C.c_double * 4)()
        self._lib.WelsHipGroupHostStats(self._h, out)
        return dict(stage_ms_per_picture=out[0], entropy_ms_per_picture=out[1], pictures=int(out[2]), packed_record_bytes_per_picture=out[3])

    def recon(self, session):
        n = self.w * self.h * 3 // 2
        buf = (C.c_uint8 * n)()
        rc = self._lib.WelsHipGroupGetReconFrame(self._h, session, buf, n)
        if rc:
            raise WelsHipError(rc, "recon")
        return bytes(buf)

    def backend_name(self):
        return (self._lib.WelsHipGroupBackendName(self._h) or b"").decode()

    def close(self):
        if self._h:
            self._lib.WelsHipGroupDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
