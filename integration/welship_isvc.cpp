// welship_isvc.cpp -- TEST INFRASTRUCTURE: the drop-in boundary made concrete.
//
// An ISVCEncoder (codec/api/wels/codec_api.h:272-343) whose methods forward to the C ABI of include/welship.h, plus the
// two factory symbols applications link against (WelsCreateSVCEncoder / WelsDestroySVCEncoder, codec_api.h:545,552).
// oracle/Makefile links it with the reference's own, unmodified console front-end (codec/console/enc/src/welsenc.cpp)
// into oracle/_ref/h264enc_welship: the reference's CLI then runs on this engine, and tests/test_dropin_cli.py checks
// that it writes the same bytes as the reference's CLI on the reference encoder.  This is INTEGRATION.md (A) compiled.
//
// The engine library is opened at run time: $WELSHIP_LIB (libwelship.so on an MI355X box, the CPU wave-emulation build
// in the CPU-only test tier).  Needs the reference's API headers at build time only; nothing from the reference is copied.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "codec_api.h"
#include "../include/welship.h"

namespace {

struct Api {
  void* so = nullptr;
  int (*Create) (WelsHipEncoder**) = nullptr;
  void (*Destroy) (WelsHipEncoder*) = nullptr;
  int (*GetDefaultParams) (WelsHipEncoder*, WelsHipEncParam*) = nullptr;
  int (*InitializeExt) (WelsHipEncoder*, const WelsHipEncParam*) = nullptr;
  int (*Uninitialize) (WelsHipEncoder*) = nullptr;
  int (*EncodeFrame) (WelsHipEncoder*, const WelsHipSourcePicture*, WelsHipFrameBSInfo*) = nullptr;
  int (*ForceIntraFrame) (WelsHipEncoder*, int) = nullptr;
  int (*GetReconFrame) (WelsHipEncoder*, uint8_t*, size_t) = nullptr;
  int (*SetOption) (WelsHipEncoder*, int, void*) = nullptr;
  int (*EncodeParameterSets) (WelsHipEncoder*, WelsHipFrameBSInfo*) = nullptr;
  int (*GetOption) (WelsHipEncoder*, int, void*) = nullptr;
  const char* (*GetLastError) (void) = nullptr;
  bool load() {
    if (so) return true;
    const char* path = getenv ("WELSHIP_LIB");
    so = dlopen (path && *path ? path : "libwelship.so", RTLD_NOW | RTLD_LOCAL);
    if (!so) { fprintf (stderr, "welship_isvc: cannot open the engine library (%s): %s\n", path ? path : "libwelship.so", dlerror()); return false; }
#define SYM(field, name) field = reinterpret_cast<decltype (field)> (dlsym (so, name)); if (!field) { fprintf (stderr, "welship_isvc: missing symbol %s\n", name); return false; }
    SYM (Create, "WelsHipCreateEncoder") SYM (Destroy, "WelsHipDestroyEncoder") SYM (GetDefaultParams, "WelsHipGetDefaultParams")
    SYM (InitializeExt, "WelsHipInitializeExt") SYM (Uninitialize, "WelsHipUninitialize") SYM (EncodeFrame, "WelsHipEncodeFrame")
    SYM (ForceIntraFrame, "WelsHipForceIntraFrame") SYM (GetLastError, "WelsHipGetLastError") SYM (GetReconFrame, "WelsHipGetReconFrame")
    SYM (SetOption, "WelsHipSetOption") SYM (GetOption, "WelsHipGetOption") SYM (EncodeParameterSets, "WelsHipEncodeParameterSets")
#undef SYM
    return true;
  }
} g_api;

class CWelsHipEncoder : public ISVCEncoder {
 public:
  CWelsHipEncoder() { if (g_api.load()) g_api.Create (&m_p); }
  virtual ~CWelsHipEncoder() { if (m_p) g_api.Destroy (m_p); }
  bool Ok() const { return m_p != NULL; }

  virtual int EXTAPI Initialize (const SEncParamBase* b) {
    if (!b) return cmInitParaError;
    SEncParamExt e;
    GetDefaultParams (&e);
    e.iUsageType = b->iUsageType; e.iPicWidth = b->iPicWidth; e.iPicHeight = b->iPicHeight;
    e.iTargetBitrate = b->iTargetBitrate; e.iRCMode = b->iRCMode; e.fMaxFrameRate = b->fMaxFrameRate;
    e.sSpatialLayers[0].iVideoWidth = b->iPicWidth; e.sSpatialLayers[0].iVideoHeight = b->iPicHeight;
    e.sSpatialLayers[0].fFrameRate = b->fMaxFrameRate; e.sSpatialLayers[0].iSpatialBitrate = b->iTargetBitrate;
    return InitializeExt (&e);
  }
  virtual int EXTAPI InitializeExt (const SEncParamExt* p) {
    if (!p || !m_p) return cmInitParaError;
    WelsHipEncParam q;
    g_api.GetDefaultParams (m_p, &q);
    q.iUsageType = p->iUsageType; q.iPicWidth = p->iPicWidth; q.iPicHeight = p->iPicHeight;
    // the stream's level follows the layer's bitrate (au_set.cpp:526 reads sSpatialLayers[].iSpatialBitrate)
    q.iTargetBitrate = p->sSpatialLayers[0].iSpatialBitrate > 0 ? p->sSpatialLayers[0].iSpatialBitrate : p->iTargetBitrate;
    q.iRCMode = p->iRCMode; q.fMaxFrameRate = p->fMaxFrameRate;
    q.iTemporalLayerNum = p->iTemporalLayerNum; q.iSpatialLayerNum = p->iSpatialLayerNum;
    q.iComplexityMode = p->iComplexityMode; q.uiIntraPeriod = p->uiIntraPeriod;
    q.eSpsPpsIdStrategy = p->eSpsPpsIdStrategy; q.iEntropyCodingModeFlag = p->iEntropyCodingModeFlag;
    q.iLoopFilterDisableIdc = p->iLoopFilterDisableIdc; q.iLoopFilterAlphaC0Offset = p->iLoopFilterAlphaC0Offset;
    q.iLoopFilterBetaOffset = p->iLoopFilterBetaOffset; q.bEnableFrameCroppingFlag = p->bEnableFrameCroppingFlag;
    const SSpatialLayerConfig& l = p->sSpatialLayers[0];
    q.iDLayerQp = l.iDLayerQp;
    q.uiSliceMode = l.sSliceArgument.uiSliceMode; q.uiSliceNum = l.sSliceArgument.uiSliceNum;
    for (int k = 0; k < MAX_SLICES_NUM_TMP && k < 35; ++k) q.uiSliceMbNum[k] = l.sSliceArgument.uiSliceMbNum[k];
    q.bEnableAdaptiveQuant = p->bEnableAdaptiveQuant; q.bEnableBackgroundDetection = p->bEnableBackgroundDetection;
    q.bEnableSceneChangeDetect = p->bEnableSceneChangeDetect; q.bEnableLongTermReference = p->bEnableLongTermReference;
    q.bEnableDenoise = p->bEnableDenoise; q.bEnableFrameSkip = p->bEnableFrameSkip;
    q.iMultipleThreadIdc = p->iMultipleThreadIdc;
    const int rc = g_api.InitializeExt (m_p, &q);
    if (rc) fprintf (stderr, "welship_isvc: InitializeExt: %s\n", g_api.GetLastError());
    m_w = p->iPicWidth; m_h = p->iPicHeight; m_frames = 0;
    return rc;
  }
  virtual int EXTAPI GetDefaultParams (SEncParamExt* p) {
    // the documented defaults (param_svc.h:132-211 FillDefault) of the fields this engine reads; the caller's struct is
    // otherwise zeroed, exactly what an application sees before it fills in its own values
    if (!p) return cmInitParaError;
    memset (p, 0, sizeof (*p));
    WelsHipEncParam q;
    if (m_p) g_api.GetDefaultParams (m_p, &q); else memset (&q, 0, sizeof (q));
    p->iUsageType = (EUsageType)q.iUsageType; p->iRCMode = (RC_MODES)q.iRCMode; p->fMaxFrameRate = q.fMaxFrameRate;
    p->iTemporalLayerNum = q.iTemporalLayerNum; p->iSpatialLayerNum = q.iSpatialLayerNum;
    p->iComplexityMode = (ECOMPLEXITY_MODE)q.iComplexityMode; p->uiIntraPeriod = q.uiIntraPeriod;
    p->eSpsPpsIdStrategy = (EParameterSetStrategy)q.eSpsPpsIdStrategy; p->iEntropyCodingModeFlag = q.iEntropyCodingModeFlag;
    p->iLoopFilterDisableIdc = q.iLoopFilterDisableIdc; p->bEnableFrameCroppingFlag = q.bEnableFrameCroppingFlag != 0;
    p->iMultipleThreadIdc = 1; p->iNumRefFrame = 1; p->iMaxQp = 51; p->iMinQp = 0;
    p->sSpatialLayers[0].iDLayerQp = q.iDLayerQp; p->sSpatialLayers[0].uiProfileIdc = PRO_BASELINE;
    p->sSpatialLayers[0].sSliceArgument.uiSliceMode = SM_SINGLE_SLICE; p->sSpatialLayers[0].sSliceArgument.uiSliceNum = 1;
    return cmResultSuccess;
  }
  virtual int EXTAPI Uninitialize() { return m_p ? g_api.Uninitialize (m_p) : cmInitParaError; }
  virtual int EXTAPI EncodeFrame (const SSourcePicture* s, SFrameBSInfo* o) {
    if (!s || !o || !m_p) return cmInitParaError;
    WelsHipSourcePicture sp;
    memset (&sp, 0, sizeof (sp));
    sp.iColorFormat = s->iColorFormat;
    for (int k = 0; k < 4; ++k) { sp.iStride[k] = s->iStride[k]; sp.pData[k] = s->pData[k]; }
    sp.iPicWidth = s->iPicWidth; sp.iPicHeight = s->iPicHeight; sp.uiTimeStamp = s->uiTimeStamp;
    WelsHipFrameBSInfo b;
    const int rc = g_api.EncodeFrame (m_p, &sp, &b);
    if (rc) { fprintf (stderr, "welship_isvc: EncodeFrame: %s\n", g_api.GetLastError()); return rc; }
    memset (o, 0, sizeof (*o));
    o->iLayerNum = b.iLayerNum; o->eFrameType = (EVideoFrameType)b.eFrameType;
    o->iFrameSizeInBytes = b.iFrameSizeInBytes; o->uiTimeStamp = b.uiTimeStamp;
    for (int i = 0; i < b.iLayerNum; ++i) {         // same ownership rule: the buffers stay valid until the next EncodeFrame
      o->sLayerInfo[i].uiLayerType = b.sLayerInfo[i].uiLayerType; o->sLayerInfo[i].eFrameType = (EVideoFrameType)b.sLayerInfo[i].eFrameType;
      o->sLayerInfo[i].uiTemporalId = b.sLayerInfo[i].uiTemporalId; o->sLayerInfo[i].uiSpatialId = b.sLayerInfo[i].uiSpatialId;
      o->sLayerInfo[i].uiQualityId = b.sLayerInfo[i].uiQualityId; o->sLayerInfo[i].iSubSeqId = b.sLayerInfo[i].iSubSeqId;
      o->sLayerInfo[i].iNalCount = b.sLayerInfo[i].iNalCount; o->sLayerInfo[i].pNalLengthInByte = b.sLayerInfo[i].pNalLengthInByte;
      o->sLayerInfo[i].pBsBuf = b.sLayerInfo[i].pBsBuf;
    }
    if (!m_dump.empty() && b.eFrameType != WelsHipFrameTypeSkip) {   // ENCODER_OPTION_DUMP_FILE: the reconstructed pictures, appended
      std::vector<uint8_t> rec ((size_t)m_w * m_h * 3 / 2);
      if (g_api.GetReconFrame (m_p, rec.data(), rec.size()) == 0) {
        FILE* f = fopen (m_dump.c_str(), m_frames == 0 ? "wb" : "ab");
        if (f) { fwrite (rec.data(), 1, rec.size(), f); fclose (f); }
      }
    }
    ++m_frames;
    return cmResultSuccess;
  }
  virtual int EXTAPI EncodeParameterSets (SFrameBSInfo* o) {
    if (!o || !m_p) return cmInitParaError;
    WelsHipFrameBSInfo b;
    const int rc = g_api.EncodeParameterSets (m_p, &b);
    if (rc) return rc;
    memset (o, 0, sizeof (*o));
    o->iLayerNum = 1; o->eFrameType = videoFrameTypeInvalid; o->iFrameSizeInBytes = b.iFrameSizeInBytes;
    o->sLayerInfo[0].uiLayerType = NON_VIDEO_CODING_LAYER; o->sLayerInfo[0].eFrameType = videoFrameTypeInvalid;
    o->sLayerInfo[0].iNalCount = b.sLayerInfo[0].iNalCount; o->sLayerInfo[0].pNalLengthInByte = b.sLayerInfo[0].pNalLengthInByte;
    o->sLayerInfo[0].pBsBuf = b.sLayerInfo[0].pBsBuf;
    return cmResultSuccess;
  }
  virtual int EXTAPI ForceIntraFrame (bool bIdr, int = -1) { return m_p ? g_api.ForceIntraFrame (m_p, bIdr ? 1 : 0) : cmInitParaError; }
  virtual int EXTAPI SetOption (ENCODER_OPTION id, void* pOption) {
    if (id == ENCODER_OPTION_DUMP_FILE && pOption) {
      const SDumpLayer* d = static_cast<const SDumpLayer*> (pOption);
      if (d->iLayer != 0 || !d->pFileName) return cmInitParaError;
      m_dump = d->pFileName;
      return cmResultSuccess;
    }
    if (id == ENCODER_OPTION_TRACE_LEVEL || id == ENCODER_OPTION_TRACE_CALLBACK || id == ENCODER_OPTION_TRACE_CALLBACK_CONTEXT) return cmResultSuccess;
    return m_p ? g_api.SetOption (m_p, (int)id, pOption) : cmInitExpected;     // the enum values are the C ABI's ids
  }
  virtual int EXTAPI GetOption (ENCODER_OPTION id, void* pOption) { return m_p ? g_api.GetOption (m_p, (int)id, pOption) : cmInitExpected; }

 private:
  WelsHipEncoder* m_p = NULL;
  std::string m_dump;
  int m_w = 0, m_h = 0, m_frames = 0;
};

}  // namespace

extern "C" int WelsCreateSVCEncoder (ISVCEncoder** ppEncoder) {
  if (!ppEncoder) return 1;
  CWelsHipEncoder* e = new CWelsHipEncoder();
  if (!e->Ok()) { delete e; *ppEncoder = NULL; return 1; }
  *ppEncoder = e;
  return 0;
}
extern "C" void WelsDestroySVCEncoder (ISVCEncoder* pEncoder) { delete static_cast<CWelsHipEncoder*> (pEncoder); }
