// welship_hooks.cpp -- the reference-side binding of libwelship.so at the SWelsFuncPtrList dispatch surface.
//
// This file is compiled INTO a patched copy of the reference encoder (integration/openh264_hip.patch, HAVE_HIP), against
// the reference's own headers, exactly like an x86/NEON back end would be: it is the code an OpenH264 maintainer adds.  It is
// not part of libwelship.so and it contains no mode decision: it converts between the reference's structures and the C ABI of
// include/welship.h (2b, the explicit frame API) and keeps the reference's frame layer, reference-list management, pre-
// processing, rate control and entropy coder in charge.
//
//   WelsHipInstall (pFuncList, pParam)       end of InitFunctionPointers (codec/encoder/core/src/encoder.cpp:157-232): decides
//                                            whether this session can run on the device and fills the three hook pointers
//   pfHipFrameMd (pCtx)                      after PreprocessSliceCoding (encoder_ext.cpp:3652): the whole picture on the
//                                            device -- WelsISliceMdEnc / WelsMdInterMbLoop (svc_encode_slice.cpp:534-599,
//                                            1807-1899) for every slice, PerformDeblockingFilter (deblocking.cpp:744-762),
//                                            border expansion (ref_list_mgr_svc.cpp:375)
//   pfHipCodeSlice (pCtx, pSlice)            in WelsCodeOneSlice (svc_encode_slice.cpp:1642) instead of g_pWelsSliceCoding[][]:
//                                            the host loop that is left -- pfWelsRcMbInit, neighbour caches,
//                                            pfWelsSpatialWriteMbSyn, pfWelsRcMbInfoUpdate per macroblock
//   pfHipRelease (state)                     WelsUninitEncoderExt (encoder_ext.cpp:2239)
//   pfHipVaaCalc (pCtx, did, cur, ref ..)    CWelsPreProcess::AnalyzeSpatialPic (wels_preprocess.cpp:263-310): VaaCalculation's statistics of the layer's
//                                            source picture (8x8 SADs, sums of differences, variances) on the device
//   pfHipDownsample (state, dst, src ..)     CWelsPreProcess::DownsamplePadding (wels_preprocess.cpp:625-675): a spatial layer's source picture from
//                                            the next larger one -- the down-sampling cascade of codec/processing on the device
//
//   WELS_HIP_LEAVES=1                        instead of all of the above: the reference's own loops stay in charge and every LEAF slot of the table
//                                            (SAD / SATD, transforms, quantisation, motion compensation, the 28 intra predictors, the edge filters)
//                                            is pointed at the device-backed export of the same typedef (include/welship_leaf.h) -- InstallLeaves
//                                            below; a launch per call, for bring-up and parity checking only
//
// What runs on the device (see WelsHipSupported below): camera video and screen content, CAVLC and CABAC, slice threads, temporal layers,
// LTR, denoising, scene-change and background detection, frame skipping, all rate-control modes.  With a frame-constant QP
// (rate control off, or on with more than one slice, or I pictures in bitrate mode: WelsRcMbInitGom with bEnableGomQp ==
// false, ratectl.cpp:1199-1204,1239-1262) a picture is one device call; with GOM-level QP (one slice per picture) the QP of a
// group of macroblocks depends on the bits the groups before it produced, so the picture is coded group by group from inside
// the slice loop -- a latency chain of one device call per group, bit-exact but not the throughput path -- unless the picture is a
// P picture of a camera session with whole-row groups: then the recursion runs inside the kernel and the picture is one call
// (WELS_HIP_GOM, default 2).  Size-limited slices: the device codes ahead of the entropy writer (WELS_HIP_DYNSLICE, default on).
// What keeps the reference's C path: SVC inter-layer prediction (and whatever the switches above take away) -- the hooks stay
// NULL, as they would on a CPU without the needed SIMD level.
#if defined(HAVE_HIP)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <chrono>
#include <map>
#include <mutex>
#include <atomic>
#include <vector>

#include "encoder_context.h"
#include "svc_enc_slice_segment.h"
#include "svc_encode_slice.h"
#include "svc_base_layer_md.h"
#include "svc_motion_estimate.h"
#include "svc_mode_decision.h"
#include "deblocking.h"
#include "svc_set_mb_syn.h"
#include "svc_enc_golomb.h"
#include "rc.h"
#include "wels_preprocess.h"
#include <dlfcn.h>
#include "compact.h"
#include "welship.h"
#include "welship_leaf.h"
#include <type_traits>
#include "wh_types.h"          // WhMbRecord: the engine's per-macroblock record (openh264_amd/csrc/common)

namespace WelsEnc {

// defined in svc_mode_decision.cpp:520-537 (no header declares it): what WelsInitSCDPskipFunc installs for screen content below HIGH complexity
bool WelsMdInterJudgeSCDPskip (sWelsEncCtx* pEncCtx, SWelsMD* pWelsMd, SSlice* slice, SMB* pCurMb, SMbCache* pMbCache);
// defined in ratectl.cpp:689-709 (no header declares it): the layer whose per-group SADs RcGomTargetBits reads, or NULL for this layer's own
SWelsSvcRc* RcJudgeBaseUsability (sWelsEncCtx* pEncCtx);

namespace {

// libwelship.so is bound at run time ($WELSHIP_LIB, else the name the loader finds): a build of the reference with HAVE_HIP
// still runs -- on its C path -- on a machine without the library or without an MI355X.
struct HipApi {
  int (*FrameCtxCreate) (WelsHipFrameCtx**, const WelsHipFrameCfg*);
  void (*FrameCtxDestroy) (WelsHipFrameCtx*);
  int (*FrameEncode) (WelsHipFrameCtx*, const WelsHipFrameJob*, const void**);
  int (*FrameGetPicture) (WelsHipFrameCtx*, int, uint8_t* const*, const int32_t*);
  int (*FrameGetMbStates) (WelsHipFrameCtx*, int, void*, size_t);
  int (*DownsamplePicture) (int, uint8_t* const*, const int32_t*, int32_t, int32_t, const uint8_t* const*, const int32_t*, int32_t, int32_t);      // optional
  int (*FrameVaa) (WelsHipFrameCtx*, const WelsHipVaaJob*);
  int (*FrameBgd) (WelsHipFrameCtx*, const WelsHipBgdJob*);                                                                                         // optional
  const char* (*GetLastError) (void);
  bool ok;
};
HipApi g_api = { NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, false };
bool LoadApi() {
  static std::mutex mu;                  // several encoders of one process may be initialised at once
  std::lock_guard<std::mutex> lock (mu);
  if (g_api.ok) return true;
  const char* path = getenv ("WELSHIP_LIB");
  void* h = dlopen (path && *path ? path : "libwelship.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) return false;
  g_api.FrameCtxCreate = (int (*) (WelsHipFrameCtx**, const WelsHipFrameCfg*))dlsym (h, "WelsHipFrameCtxCreate");
  g_api.FrameCtxDestroy = (void (*) (WelsHipFrameCtx*))dlsym (h, "WelsHipFrameCtxDestroy");
  g_api.FrameEncode = (int (*) (WelsHipFrameCtx*, const WelsHipFrameJob*, const void**))dlsym (h, "WelsHipFrameEncode");
  g_api.FrameGetPicture = (int (*) (WelsHipFrameCtx*, int, uint8_t* const*, const int32_t*))dlsym (h, "WelsHipFrameGetPicture");
  g_api.FrameGetMbStates = (int (*) (WelsHipFrameCtx*, int, void*, size_t))dlsym (h, "WelsHipFrameGetMbStates");
  g_api.GetLastError = (const char* (*) (void))dlsym (h, "WelsHipGetLastError");
  g_api.FrameVaa = (int (*) (WelsHipFrameCtx*, const WelsHipVaaJob*))dlsym (h, "WelsHipFrameVaa");
  g_api.FrameBgd = (int (*) (WelsHipFrameCtx*, const WelsHipBgdJob*))dlsym (h, "WelsHipFrameBgd");
  g_api.DownsamplePicture = (int (*) (int, uint8_t* const*, const int32_t*, int32_t, int32_t, const uint8_t* const*, const int32_t*, int32_t, int32_t))dlsym (h, "WelsHipDownsamplePicture");
  g_api.ok = g_api.FrameCtxCreate && g_api.FrameCtxDestroy && g_api.FrameEncode && g_api.FrameGetPicture && g_api.FrameGetMbStates && g_api.GetLastError;
  return g_api.ok;
}

struct HipLayer {                       // one spatial layer = one device context
  WelsHipFrameCtx* ctx = NULL;
  std::map<const SPicture*, int> twin;   // reference SPicture -> device picture index
  int num_pictures = 0;
  const WhMbRecord* records = NULL;      // of the picture being coded (valid from pfHipFrameMd to the end of its slices)
  const WelsHipPackedRecords* packed = NULL;   // ... or its packed records (whole-picture calls: WelsHipFrameJob::bPackedRecords); a macroblock is
                                               // expanded into a WhMbRecord of the slice loop's own when the entropy writer gets to it
  // GOM-level rate control (one slice per picture, QP per group of macroblocks from the bits written so far): the picture is
  // coded group by group from inside the slice loop -- the job is only prepared by pfHipFrameMd
  bool gom = false;
  WelsHipFrameJob job;
  std::vector<int32_t> first;
  std::vector<uint8_t> mb_qp;
  int coded_upto = 0;
  // size-limited slices (SM_SIZELIMITED_SLICE): the device codes AHEAD of the entropy writer -- from inside the slice loop, the rest of
  // the picture as if the slice that begins there never ended; where the writer ends the slice, the next one begins with another call
  // (WelsHipFrameJob::iDynSlice).  The picture-wide passes follow the picture's last macroblock.
  // With several slice threads the picture is split into one PARTITION per thread (rows of macroblocks), each sliced on its own by its
  // thread's task (CWelsConstrainedSizeSlicingEncodingTask, wels_task_encoder.cpp:231-325); a partition boundary is a slice boundary.
  bool dyn = false;
  struct DynPart {
    int coded_upto = 0;                  // the records of [.., coded_upto) of this partition are valid ...
    int coded_slice = -1;                //   ... as macroblocks of this slice
    int est[2] = {0, 0};                 // macroblocks per slice lately (I / P pictures): how far ahead of the writer a call codes
  } part[MAX_THREADS_NUM];
  std::mutex dyn_mu;                     // the counters below (slice tasks of a picture run concurrently)
  int dyn_calls = 0, dyn_slices = 0, dyn_parts_left = 0;
  long dyn_mbs_coded = 0;                // macroblocks the device coded for the picture, the ones coded in vain (ahead of a slice end) included
  WelsHipGomRc gomrc;                    // GOM-level rate control inside the kernel (WELS_HIP_GOM=2): the picture's rate-control inputs
  WelsHipScreenInfo screen;              // screen content: the pre-processing's results of the picture being coded
  std::vector<uint32_t> fme_down;        //   and what the device reports back per slice (uiSliceFMECostDown)
  std::vector<int16_t> il_hint;          // highest layer of a multi-layer session: hints from the layer below
  std::vector<WhMbState> states;         // lower layers of a multi-layer session: the device's motion data, for the layer above
  // WELS_HIP_CHECK_VAA=1: what the device computed for this picture's pre-analysis, compared in HipFrameMd with what the reference's own
  // C functions then left in pVaa (the hook lets them run as well)
  std::vector<int8_t> bgd_copy;          // WELS_HIP_CHECK_BGD=1: the device's flags of the picture about to be coded
  struct VaaCopy { bool valid = false; int n = 0, bgd = 0, var = 0, ssd = 0, frame_sad = 0; std::vector<int32_t> sad, sd, sum, sq, ssd16; std::vector<uint8_t> mad; } vaa_copy;
  std::vector<WelsHipMbReencode> reencode;   // macroblocks of the picture being coded that were coded again at a higher QP (TRY_REENCODING)
};

struct HipState {
  HipLayer layer[MAX_DEPENDENCY_LAYER];
  int device = 0;
  std::atomic<bool> failed {false};     // a device call failed: the session reports errors from then on (slice tasks set it concurrently)
  bool trace = false;
  int layer_devices = 0;                // WELS_HIP_LAYER_DEVICES: 0 all layers on one GPU; 1 layer d on GPU base + d; n >= 2 layer d on GPU base + d mod n
  // Spatial layers with fewer macroblocks than this stay on the reference's own path (pfHipLayerOnDevice): a session's layers are coded one after the
  // other, so every layer on the device adds a whole round trip -- a latency chain of ~50 macroblock steps for 320x180, behind whatever
  // other sessions have queued -- to the frame, and a core codes 240 macroblocks in half a millisecond.  Sessions with ONE spatial layer are never
  // split this way (the caller chose the device for that size).  WELS_HIP_MIN_LAYER_MBS overrides (0: every layer on the device).
  int min_layer_mbs = 0;
  long host_pictures = 0;               // pictures of such layers (the trace's count of device pictures does not include them)
  // WELS_HIP_TRACE=2: where a picture's time goes (seconds, summed): device call incl. transfers, reconstruction copy-back,
  // entropy coding from the records
  bool timing = false;
  bool vaa_check = false;               // WELS_HIP_CHECK_VAA=1
  bool bgd = true;                      // background detection on the device (WELS_HIP_BGD=0: the VP library's function)
  int last_vaa_did = -1;                // the layer HipVaaCalc has just served: HipBgd follows it for the same picture pair
  long bgd_done = 0;
  bool bgd_check = false;               // WELS_HIP_CHECK_BGD=1: the VP library's function runs as well and HipFrameMd compares the flags
  bool vaa = true;                      // WELS_HIP_VAA=0: the pre-analysis statistics (VaaCalculation) stay the reference's C functions
  long vaa_done = 0;
  bool downsample = true;               // WELS_HIP_DOWNSAMPLE=0: the spatial layers are down-sampled by the reference's own C functions
  long downsampled = 0;
  bool gom_kernel = false;              // WELS_HIP_GOM=2: single-slice rate-controlled P pictures in ONE device call (the QP recursion runs in the kernel)
  bool packed = true;                   // whole pictures' records come back packed (where the library can: WelsHipFrameJob::pbRecordsPacked says)
  bool eager_recon = false;             // screen content: copy every reconstruction back into pDecPic as soon as the picture is coded (else: pfHipFetchRecon, on demand)
  long recon_fetched = 0;
  bool check_bits = false;              // WELS_HIP_CHECK_BITS=1: the device counts every macroblock's CAVLC bits and the slice loop compares them with the writer
  std::atomic<long> bits_checked {0};
  double t_encode = 0.0, t_getpic = 0.0, t_code = 0.0, t_vaa = 0.0, t_down = 0.0, t_md_prep = 0.0;
  int pictures = 0;
};
struct Stopwatch {
  double* acc; std::chrono::steady_clock::time_point t0;
  explicit Stopwatch (double* a) : acc (a), t0 (std::chrono::steady_clock::now()) {}
  ~Stopwatch() { if (acc) *acc += std::chrono::duration<double> (std::chrono::steady_clock::now() - t0).count(); }
};

// Is spatial layer `did` coded on the device?  (Simulcast layers are independent streams, so the answer may differ from layer to layer; the one
// coupling -- the highest layer's inter-layer hints from the layer below, HipFrameMd -- reads SMB fields that either path fills in.)
bool LayerOnDevice (const HipState* st, const sWelsEncCtx* pCtx, int did) {
  if (st->min_layer_mbs <= 0) return true;
  const SDqLayer* pLayer = pCtx->ppDqLayerList[did];
  return pLayer == NULL || pLayer->iMbWidth * pLayer->iMbHeight >= st->min_layer_mbs;
}
bool HipLayerOnDevice (sWelsEncCtx* pCtx) {
  const HipState* st = (const HipState*)pCtx->pFuncList->pHipState;
  return st == NULL || LayerOnDevice (st, pCtx, pCtx->uiDependencyId);
}

// The device's reconstruction of the layer's current picture into pDecPic.  Nothing of the hooked encoder reads pDecPic's samples on the
// host (mode decision, the in-loop filter and the reference pictures live on the device), so this only runs where the reference is about
// to: PSNR and the test builds' frame dump (pfHipFetchRecon, called from WelsEncoderEncodeExt), or for every picture of a screen-content session.
bool FetchRecon (HipState* st, sWelsEncCtx* pCtx, HipLayer& L, int iPic) {
  uint8_t* dst[3] = { pCtx->pDecPic->pData[0], pCtx->pDecPic->pData[1], pCtx->pDecPic->pData[2] };
  const int32_t ds[3] = { pCtx->pDecPic->iLineSize[0], pCtx->pDecPic->iLineSize[1], pCtx->pDecPic->iLineSize[2] };
  Stopwatch sw (st->timing ? &st->t_getpic : NULL);
  if (g_api.FrameGetPicture (L.ctx, iPic, dst, ds)) { st->failed = true; return false; }
  ++st->recon_fetched;
  return true;
}
void HipFetchRecon (sWelsEncCtx* pCtx) {
  HipState* st = (HipState*)pCtx->pFuncList->pHipState;
  if (st == NULL || st->failed || st->eager_recon) return;
  HipLayer& L = st->layer[pCtx->uiDependencyId];
  if (L.ctx != NULL) FetchRecon (st, pCtx, L, L.job.iCurPic);
}

int TwinOf (HipLayer& L, const SPicture* p) {
  std::map<const SPicture*, int>::iterator it = L.twin.find (p);
  if (it != L.twin.end()) return it->second;
  const int idx = (int)L.twin.size();
  if (idx >= L.num_pictures) return -1;
  L.twin[p] = idx;
  return idx;
}

// Frame-constant QP?  (ratectl.cpp:1199-1204: GOM-level QP only for single-slice pictures, and not for I pictures in bitrate mode)
bool FrameConstantQp (const sWelsEncCtx* pCtx) {
  const SWelsSvcCodingParam* p = pCtx->pSvcParam;
  // RC_OFF_MODE and RC_BUFFERBASED_MODE install WelsRcMbInitDisable (ratectl.cpp:1495-1516): every macroblock gets iGlobalQp (adaptive quantisation is
  // switched off by ParamValidation); the other modes run WelsRcMbInitGom, which does the same unless bEnableGomQp (ratectl.cpp:1239-1262)
  if (p->iRCMode == RC_OFF_MODE || p->iRCMode == RC_BUFFERBASED_MODE) return true;
  return !pCtx->pWelsSvcRc[pCtx->uiDependencyId].bEnableGomQp;
}

// Size-limited slices: the device codes [iFrom, ...) of partition iPart as macroblocks of slice iSliceIdx, which begins at iSliceFirst --
// about one and a half slices' worth (what the last slices of this picture type were long in this partition), the rest of the
// partition when nothing is known yet.  L.job (prepared by HipFrameMd) is read-only here: slice tasks call this concurrently.
int32_t DynCode (HipState* st, HipLayer& L, int iPart, int iSliceIdx, int iSliceFirst, int iFrom, bool is_p, int iPartFirst, int iPartEnd) {
  HipLayer::DynPart& P = L.part[iPart];
  const int est = P.est[is_p ? 1 : 0];
  int end = est > 0 ? iFrom + WELS_MAX (est + (est >> 1), 16) : iPartEnd;
  if (iFrom > iSliceFirst && end < iFrom + (iFrom - iSliceFirst)) end = iFrom + (iFrom - iSliceFirst);      // the slice outgrew the estimate: as much again as it has so far
  if (end > iPartEnd) end = iPartEnd;
  WelsHipFrameJob jb = L.job;
  jb.pSliceFirstMb = &L.first[0];
  jb.iMbBegin = iFrom; jb.iMbEnd = end;
  jb.iDynSlice = iSliceIdx + 1; jb.iDynSliceFirstMb = iSliceFirst;
  jb.bDynRedoFirst = (iFrom == iSliceFirst && iSliceFirst != iPartFirst) ? 1 : 0;      // a slice inside a partition begins with the macroblock the writer took back
  // macroblocks coded again after a CAVLC overflow (TRY_REENCODING in HipCodeSlice): every later call of the picture carries them.  A
  // macroblock a slice BEGINS with is decided from scratch (pfWelsRcMbInit and the Init functions run again in the reference): what was
  // noted for it, or for anything behind it, belongs to the slice before
  std::vector<WelsHipMbReencode> list;
  {
    std::lock_guard<std::mutex> lock (L.dyn_mu);
    if (iFrom == iSliceFirst)
      for (size_t i = 0; i < L.reencode.size();) { if (L.reencode[i].iMbXY >= iFrom && L.reencode[i].iMbXY < iPartEnd) L.reencode.erase (L.reencode.begin() + i); else ++i; }
    list = L.reencode;
  }
  if (!list.empty()) { jb.pReencode = &list[0]; jb.iNumReencode = (int32_t)list.size(); }
  const void* rec = NULL;
  const int rc = g_api.FrameEncode (L.ctx, &jb, &rec);
  if (rc) { fprintf (stderr, "welship hooks: WelsHipFrameEncode (slice %d from MB %d) failed (%d: %s)\n", iSliceIdx, iFrom, rc, g_api.GetLastError()); st->failed = true; return ENC_RETURN_UNEXPECTED; }
  L.records = (const WhMbRecord*)rec; L.packed = NULL;          // (the context's record array: the same pointer for every call)
  P.coded_upto = end; P.coded_slice = iSliceIdx;
  std::lock_guard<std::mutex> lock (L.dyn_mu);
  ++L.dyn_calls;
  L.dyn_mbs_coded += end - iFrom;
  return ENC_RETURN_SUCCESS;
}

// the device context of spatial layer `did` (created with the layer's first call: pre-analysis or mode decision)
bool EnsureLayerCtx (HipState* st, sWelsEncCtx* pCtx, int did) {
  HipLayer& L = st->layer[did];
  if (L.ctx != NULL) return true;
  const SWelsSvcCodingParam* pParam = pCtx->pSvcParam;
  const SDqLayer* pLayer = pCtx->ppDqLayerList[did];
  WelsHipFrameCfg cfg;
  memset (&cfg, 0, sizeof (cfg));
  cfg.iDevice = st->device + (st->layer_devices == 1 ? did : st->layer_devices >= 2 ? did % st->layer_devices : 0);
  cfg.iPicWidth = pLayer->iMbWidth * 16; cfg.iPicHeight = pLayer->iMbHeight * 16;
  cfg.iNumPictures = WELS_MAX (pParam->iNumRefFrame, pParam->iMaxNumRefFrame) + 2;    // RequestMemorySvc allocates 1 + iMaxNumRefFrame pictures per layer
  L.num_pictures = cfg.iNumPictures;
  const int rc = g_api.FrameCtxCreate (&L.ctx, &cfg);
  if (rc) { fprintf (stderr, "welship hooks: no device context (%d: %s)\n", rc, g_api.GetLastError()); st->failed = true; return false; }
  if (st->trace) fprintf (stderr, "welship hooks: device context of layer %d created (%dx%d, %d pictures)\n", did, cfg.iPicWidth, cfg.iPicHeight, cfg.iNumPictures);
  return true;
}

// CWelsPreProcess::VaaCalculation (wels_preprocess.cpp:677-711) for camera video: the statistics of the layer's source picture against
// the source picture of its reference -- 8x8 SADs (what LOW-complexity mode decision and the rate control's complexity analysis read),
// and with background detection / the variance request also sums of differences, largest differences, sums and sums of squares
// (vaacalcfuncs.cpp) -- as one call into libwelship.so.  The current picture stays on the device for HipFrameMd; the other one is there
// from the picture before.  BackgroundDetection and the complexity analysis that follow read the arrays this fills.  Non-zero: not done.
int32_t HipVaaCalc (sWelsEncCtx* pCtx, int32_t iDid, SPicture* pCurPic, SPicture* pRefPic, bool bCalculateSQDiff, bool bCalculateVar, bool bCalculateBGD) {
  HipState* st = (HipState*)pCtx->pFuncList->pHipState;
  if (st != NULL) st->last_vaa_did = -1;        // (set again only when the device really served this call: HipBgd must never address a stale layer)
  if (st == NULL || !st->vaa || st->failed || g_api.FrameVaa == NULL || pCtx->pVaa == NULL || pCurPic == NULL || pRefPic == NULL) return 1;
  if (iDid < 0 || iDid >= MAX_DEPENDENCY_LAYER || pCurPic->pData[0] == NULL || pRefPic->pData[0] == NULL || pCurPic->pData[0] == pRefPic->pData[0]) return 1;
  if (pCurPic->iLineSize[0] != pRefPic->iLineSize[0]) return 1;               // (the C functions take ONE stride for both pictures)
  if (!LayerOnDevice (st, pCtx, iDid)) return 1;                              // (a layer that is coded on the host: its statistics come from there too)
  if (!EnsureLayerCtx (st, pCtx, iDid)) return 1;
  SVAACalcResult* pRes = &pCtx->pVaa->sVaaCalcInfo;
  WelsHipVaaJob job;
  memset (&job, 0, sizeof (job));
  for (int i = 0; i < 3; ++i) { job.pCur[i] = pCurPic->pData[i]; job.iCurStride[i] = pCurPic->iLineSize[i]; job.pRef[i] = pRefPic->pData[i]; job.iRefStride[i] = pRefPic->iLineSize[i]; }
  job.iPicWidth = pCurPic->iWidthInPixel; job.iPicHeight = pCurPic->iHeightInPixel;
  job.bCalcVar = bCalculateVar ? 1 : 0; job.bCalcBgd = bCalculateBGD ? 1 : 0; job.bCalcSsd = bCalculateSQDiff ? 1 : 0;
  job.pSad8x8 = pRes->pSad8x8 ? &pRes->pSad8x8[0][0] : NULL;
  job.pSsd16x16 = pRes->pSsd16x16; job.pSum16x16 = pRes->pSum16x16; job.pSumOfSquare16x16 = pRes->pSumOfSquare16x16;
  job.pSumOfDiff8x8 = pRes->pSumOfDiff8x8 ? &pRes->pSumOfDiff8x8[0][0] : NULL;
  job.pMad8x8 = pRes->pMad8x8 ? &pRes->pMad8x8[0][0] : NULL;
  job.pFrameSad = &pRes->iFrameSad;
  int rc;
  { Stopwatch sw (st->timing ? &st->t_vaa : NULL); rc = g_api.FrameVaa (st->layer[iDid].ctx, &job); }
  if (rc != 0) {
    if (st->trace) fprintf (stderr, "welship hooks: pre-analysis of layer %d stays on the host (%d: %s)\n", iDid, rc, g_api.GetLastError());
    return 1;
  }
  pRes->pCurY = pCurPic->pData[0]; pRes->pRefY = pRefPic->pData[0];         // (what VaaCalculation / CVAACalculation::Process leave there)
  ++st->vaa_done;
  st->last_vaa_did = iDid;
  if (st->vaa_check) {
    HipLayer::VaaCopy& V = st->layer[iDid].vaa_copy;
    const int n = (job.iPicWidth >> 4) * (job.iPicHeight >> 4);
    const bool sd = job.bCalcBgd != 0, sum = job.bCalcSsd || (!job.bCalcBgd && job.bCalcVar), ssd = job.bCalcSsd != 0;
    V.valid = true; V.n = n; V.bgd = sd; V.var = sum; V.ssd = ssd; V.frame_sad = pRes->iFrameSad;
    V.sad.assign (job.pSad8x8, job.pSad8x8 + 4 * n);
    if (sd) { V.sd.assign (job.pSumOfDiff8x8, job.pSumOfDiff8x8 + 4 * n); V.mad.assign (job.pMad8x8, job.pMad8x8 + 4 * n); }
    if (sum) { V.sum.assign (job.pSum16x16, job.pSum16x16 + n); V.sq.assign (job.pSumOfSquare16x16, job.pSumOfSquare16x16 + n); }
    if (ssd) V.ssd16.assign (job.pSsd16x16, job.pSsd16x16 + n);
    return 1;                 // the C functions run as well; HipFrameMd compares
  }
  if (st->trace) fprintf (stderr, "welship hooks: pre-analysis statistics of layer %d on the device (bgd %d var %d ssd %d)\n", iDid, job.bCalcBgd, job.bCalcVar, job.bCalcSsd);
  return 0;
}

// CWelsPreProcess::BackgroundDetection (wels_preprocess.cpp:713-761) of the picture pair HipVaaCalc has just analysed: the in-place pass of
// BackgroundDetection.cpp:333-374 on the device, from the statistics that are still there.  Non-zero: the VP library's function runs.
int32_t HipBgd (sWelsEncCtx* pCtx, SVAAFrameInfo* pVaaInfo, SPicture* pCurPic, SPicture* pRefPic) {
  HipState* st = (HipState*)pCtx->pFuncList->pHipState;
  if (st == NULL || st->failed || !st->bgd || !g_api.FrameBgd || st->vaa_check) return 1;
  for (int i = 0; i < MAX_DEPENDENCY_LAYER; ++i) st->layer[i].bgd_copy.clear();
  const int did = st->last_vaa_did;
  st->last_vaa_did = -1;
  if (did < 0 || st->layer[did].ctx == NULL) return 1;       // (the statistics of this call's pictures came from the C functions)
  WelsHipBgdJob job;
  memset (&job, 0, sizeof (job));
  for (int i = 0; i < 3; ++i) { job.pCur[i] = pCurPic->pData[i]; job.pRef[i] = pRefPic->pData[i]; }
  job.iPicWidth = pCurPic->iWidthInPixel; job.iPicHeight = pCurPic->iHeightInPixel;
  job.pBackgroundMbFlag = pVaaInfo->pVaaBackgroundMbFlag;
  if (st->bgd_check) {         // into a copy of the caller's array (entries outside the covered area keep their values in both)
    const size_t n = (size_t) ((job.iPicWidth + 15) >> 4) * ((job.iPicHeight + 15) >> 4);
    st->layer[did].bgd_copy.assign (pVaaInfo->pVaaBackgroundMbFlag, pVaaInfo->pVaaBackgroundMbFlag + n);
    job.pBackgroundMbFlag = st->layer[did].bgd_copy.data();
  }
  const int rc = g_api.FrameBgd (st->layer[did].ctx, &job);
  if (rc != 0) {
    st->layer[did].bgd_copy.clear();
    if (st->trace) fprintf (stderr, "welship hooks: background detection of layer %d stays on the host (%d: %s)\n", did, rc, g_api.GetLastError());
    return 1;
  }
  ++st->bgd_done;
  if (st->bgd_check) return 1;             // the VP library's function runs as well; HipFrameMd compares
  if (st->trace) fprintf (stderr, "welship hooks: background detection of layer %d on the device\n", did);
  return 0;
}

int32_t HipFrameMd (sWelsEncCtx* pCtx) {
  SWelsFuncPtrList* pFunc = pCtx->pFuncList;
  HipState* st = (HipState*)pFunc->pHipState;
  if (st == NULL || st->failed) return ENC_RETURN_UNEXPECTED;
  SDqLayer* pCurLayer = pCtx->pCurDqLayer;
  const int did = pCtx->uiDependencyId;
  HipLayer& L = st->layer[did];
  const SWelsSvcCodingParam* pParam = pCtx->pSvcParam;
  const int mbw = pCurLayer->iMbWidth, mbh = pCurLayer->iMbHeight, num_mb = mbw * mbh;
  if (!LayerOnDevice (st, pCtx, did)) {          // the slice loops, the in-loop filter and the reference list run as if no hook were installed
    ++st->host_pictures;
    if (st->trace) fprintf (stderr, "welship hooks: picture of layer %d (%d macroblocks) left to the host\n", did, num_mb);
    return ENC_RETURN_SUCCESS;
  }
  if (!EnsureLayerCtx (st, pCtx, did)) return ENC_RETURN_UNEXPECTED;
  if (L.vaa_copy.valid) {      // WELS_HIP_CHECK_VAA=1: the device's pre-analysis against the reference's own
    HipLayer::VaaCopy& V = L.vaa_copy;
    V.valid = false;
    const SVAACalcResult* pRes = &pCtx->pVaa->sVaaCalcInfo;
    bool same = V.frame_sad == pRes->iFrameSad && memcmp (&V.sad[0], &pRes->pSad8x8[0][0], sizeof (int32_t) * 4 * V.n) == 0;
    if (same && V.bgd) same = memcmp (&V.sd[0], &pRes->pSumOfDiff8x8[0][0], sizeof (int32_t) * 4 * V.n) == 0 && memcmp (&V.mad[0], &pRes->pMad8x8[0][0], (size_t)4 * V.n) == 0;
    if (same && V.var) same = memcmp (&V.sum[0], pRes->pSum16x16, sizeof (int32_t) * V.n) == 0 && memcmp (&V.sq[0], pRes->pSumOfSquare16x16, sizeof (int32_t) * V.n) == 0;
    if (same && V.ssd) same = memcmp (&V.ssd16[0], pRes->pSsd16x16, sizeof (int32_t) * V.n) == 0;
    if (!same) {
      int bad = -1;
      for (int i = 0; i < 4 * V.n && bad < 0; ++i) if (V.sad[i] != (&pRes->pSad8x8[0][0])[i]) bad = i;
      fprintf (stderr, "welship hooks: the device's pre-analysis differs from the reference's (layer %d, frame SAD %d vs %d, first 8x8 SAD that differs: %d)\n", did, V.frame_sad, pRes->iFrameSad, bad);
      st->failed = true; return ENC_RETURN_UNEXPECTED;
    }
    if (st->trace) fprintf (stderr, "welship hooks: the device's pre-analysis equals the reference's\n");
  }
  if (!L.bgd_copy.empty()) {   // WELS_HIP_CHECK_BGD=1: the device's background flags against the VP library's
    const bool same = memcmp (L.bgd_copy.data(), pCtx->pVaa->pVaaBackgroundMbFlag, L.bgd_copy.size()) == 0;
    L.bgd_copy.clear();
    if (!same) { fprintf (stderr, "welship hooks: the device's background detection differs from the reference's (layer %d)\n", did); st->failed = true; return ENC_RETURN_UNEXPECTED; }
    if (st->trace) fprintf (stderr, "welship hooks: the device's background detection equals the reference's\n");
  }
  L.gom = !FrameConstantQp (pCtx);
  L.reencode.clear();
  WelsHipFrameJob& job = L.job;
  memset (&job, 0, sizeof (job));
  job.cbSize = (uint32_t)sizeof (job);
  const bool is_p = pCtx->eSliceType == P_SLICE;
  job.iCurPic = TwinOf (L, pCtx->pDecPic);
  job.iRefPic = is_p ? TwinOf (L, pCurLayer->pRefPic) : -1;
  if (job.iCurPic < 0 || (is_p && job.iRefPic < 0)) { fprintf (stderr, "welship hooks: more reference pictures than device twins\n"); st->failed = true; return ENC_RETURN_UNEXPECTED; }
  job.eSliceType = is_p ? 0 : 2;
  job.iQp = pCtx->iGlobalQp;
  job.iChromaQpIndexOffset = pCurLayer->sLayerInfo.pPpsP->uiChromaQpIndexOffset;
  job.iComplexityMode = pParam->iComplexityMode;
  job.iMvRange = pCtx->iMvRange;
  // pSlice->sScaleShift (svc_encode_slice.cpp:1652-1655), the same for every slice of the picture
  job.iMvcShift = (is_p && pCtx->uiTemporalId) ? (int) (pCtx->uiTemporalId - pCtx->pRefPic->uiTemporalId) : 0;
  std::vector<int32_t>& first = L.first;
  first.clear();
  const int nslices = GetCurrentSliceNum (pCurLayer);
  for (int i = 0; i < nslices; ++i) first.push_back (pCurLayer->pFirstMbIdxOfSlice[i]);
  first.push_back (num_mb);
  job.iNumSlices = nslices;
  job.pSliceFirstMb = &first[0];
  job.iDeblockIdc = pCurLayer->iLoopFilterDisableIdc;
  job.iAlphaOffset = pCurLayer->iLoopFilterAlphaC0Offset;
  job.iBetaOffset = pCurLayer->iLoopFilterBetaOffset;
  // the condition WelsEncoderEncodeExt evaluates after the slices are coded (encoder_ext.cpp:3870-3880); its inputs are known now
  const int iHighestTid = pParam->sDependencyLayers[did].iHighestTemporalId;
  job.bDeblock = (!pCurLayer->bDeblockingParallelFlag) && pCtx->eNalPriority != NRI_PRI_LOWEST && (iHighestTid == 0 || pCtx->uiTemporalId < iHighestTid)
                 && pCurLayer->iLoopFilterDisableIdc != 1;
  // Slice threads (iMultipleThreadIdc > 1): the reference filters slice by slice from inside its slice tasks (wels_task_encoder.cpp:184,
  // pfDeblockingFilterSlice as PreprocessSliceCoding just set it, encoder_ext.cpp:2773-2783; the filter mode is 2 then: no edge
  // crosses a slice).  That is the same picture-wide pass on the device; the host's per-slice filter is switched off for this picture.
  // (No task runs for a picture of a single slice: the reference does not filter it at all then.)
  if (pCurLayer->bDeblockingParallelFlag && pFunc->pfDeblocking.pfDeblockingFilterSlice == DeblockingFilterSliceAvcbase) {
    const SliceModeEnum eMode = pParam->sSpatialLayers[did].sSliceArgument.uiSliceMode;
    const bool tasks = eMode != SM_SINGLE_SLICE && pParam->iMultipleThreadIdc > 1;      // (size-limited slices: CWelsConstrainedSizeSlicingEncodingTask filters slice by slice too, wels_task_encoder.cpp:301)
    job.bDeblock = tasks ? 1 : 0;
    pFunc->pfDeblocking.pfDeblockingFilterSlice = DeblockingFilterSliceAvcbaseNull;
  }
  job.bExpand = pCtx->eNalPriority != NRI_PRI_LOWEST;        // UpdateRefList -> ExpandReferencingPicture (encoder_ext.cpp:3891-3899)
  for (int i = 0; i < 3; ++i) { job.pSrc[i] = pCurLayer->pEncData[i]; job.iSrcStride[i] = pCurLayer->iEncStride[i]; }
  job.pVaaSad8x8 = (is_p && pCtx->pVaa && pCtx->pVaa->sVaaCalcInfo.pSad8x8) ? &pCtx->pVaa->sVaaCalcInfo.pSad8x8[0][0] : NULL;
  job.pBgdFlags = (is_p && pParam->bEnableBackgroundDetection && pCtx->pVaa) ? pCtx->pVaa->pVaaBackgroundMbFlag : NULL;
  // WelsCodePSlice (svc_encode_slice.cpp:722-741): the highest spatial layer of a multi-layer session decides its P macroblocks
  // with WelsMdInterMbEnhancelayer -- the type and one vector of the co-located macroblock of the layer coded just before it
  // (GetRefMb / SetMvBaseEnhancelayer, svc_mode_decision.cpp:108-150), simulcast AVC included
  // pSadCost[0] lives in ONE array for all spatial layers (pEncCtx->pSadCostMb): the host's copy travels with every picture
  job.pSadCost = (pParam->iSpatialLayerNum > 1) ? pCtx->pSadCostMb : NULL;
  job.pIlHint = NULL;
  if (is_p && pCurLayer->bBaseLayerAvailableFlag && pParam->iSpatialLayerNum == did + 1 && pCurLayer->pRefLayer) {
    const SDqLayer* kpRefLayer = pCurLayer->pRefLayer;
    L.il_hint.assign ((size_t)num_mb * 4, 0);
    for (int y = 0; y < mbh; ++y)
      for (int x = 0; x < mbw; ++x) {
        const SMB* kpRefMb = &kpRefLayer->sMbDataP[ (y >> 1) * kpRefLayer->iMbWidth + (x >> 1)];
        int16_t* h = &L.il_hint[ (size_t) (y * mbw + x) * 4];
        if (IS_SVC_INTRA (kpRefMb->uiMbType)) h[2] = 1;
        else {
          const int32_t iRefMbPartIdx = ((y & 0x01) << 1) + (x & 0x01);
          const int32_t iScan4RefPartIdx = g_kuiMbCountScan4Idx[ (iRefMbPartIdx << 2)];
          h[0] = (int16_t) (kpRefMb->sMv[iScan4RefPartIdx].iMvX * (1 << 1));
          h[1] = (int16_t) (kpRefMb->sMv[iScan4RefPartIdx].iMvY * (1 << 1));
        }
      }
    job.pIlHint = &L.il_hint[0];
  }
  job.bCountBits = st->check_bits ? 1 : 0;
  job.iNumRefIdxL0Active = is_p ? pCtx->iNumRef0 : 0;
  job.pScreen = NULL;
  if (pParam->iUsageType == SCREEN_CONTENT_REAL_TIME) {
    if (!is_p) {
      // PreprocessSliceCoding (encoder_ext.cpp:2664-2671): I pictures of a screen-content session always use the SATD costs and the
      // full Intra4x4 search, whatever the complexity mode
      if (job.iComplexityMode == 0) job.iComplexityMode = 1;
    } else {
      // what the P picture's mode decision and motion estimation take from the pre-processing (scene-change / scroll detection against
      // the best reference candidate, wels_preprocess.cpp:1087-1240) and from PreprocessSliceCoding (encoder_ext.cpp:2700-2765)
      WelsHipScreenInfo& scr = L.screen;
      memset (&scr, 0, sizeof (scr));
      SVAAFrameInfoExt* pVaaExt = static_cast<SVAAFrameInfoExt*> (pCtx->pVaa);
      scr.pBlockStaticIdc = pVaaExt->pVaaBestBlockStaticIdc;
      const SPicture* pRefOri = pCurLayer->pRefOri[0];
      if (pRefOri != NULL && pRefOri->pData[1] != NULL) {
        // JudgeStaticSkip / JudgeScrollSkip address the reference's source picture with the CURRENT picture's chroma stride
        if (pRefOri->iLineSize[1] != pCurLayer->iEncStride[1]) { fprintf (stderr, "welship hooks: source pictures with different strides\n"); st->failed = true; return ENC_RETURN_UNEXPECTED; }
        scr.pRefOriChroma[0] = pRefOri->pData[1]; scr.pRefOriChroma[1] = pRefOri->pData[2]; scr.iRefOriStride = pRefOri->iLineSize[1];
      }
      scr.bStaticSkipDecision = pFunc->pfSCDPSkipDecision == WelsMdInterJudgeSCDPskip ? 1 : 0;
      scr.bScrollDetectFlag = pVaaExt->sScrollDetectInfo.bScrollDetectFlag ? 1 : 0;
      scr.iScrollMvX = pVaaExt->sScrollDetectInfo.iScrollMvX; scr.iScrollMvY = pVaaExt->sScrollDetectInfo.iScrollMvY;
      const SScreenBlockFeatureStorage* pSt = pCurLayer->pRefPic->pScreenBlockFeatureStorage;
      scr.uiSadCostThreshold16x16 = pSt ? pSt->uiSadCostThreshold[BLOCK_16x16] : 0xffffffffu;
      scr.uiSadCostThreshold8x8 = pSt ? pSt->uiSadCostThreshold[BLOCK_8x8] : 0xffffffffu;
      scr.bFeatureSearch8x8 = (pSt != NULL && pFunc->pfSearchMethod[BLOCK_8x8] == WelsDiamondCrossFeatureSearch) ? 1 : 0;
      if (scr.bFeatureSearch8x8) {
        scr.pTimesOfFeatureValue = pSt->pTimesOfFeatureValue; scr.pLocationOfFeature = pSt->pLocationOfFeature; scr.pLocationPointer = pSt->pLocationPointer;
        scr.iListSize = pSt->iActualListSize;
        const SSpatialLayerConfig& lc = pParam->sSpatialLayers[did];
        scr.iLocationEntries = (lc.iVideoWidth - 8) * (lc.iVideoHeight - 8);      // RequestScreenBlockFeatureStorage (svc_motion_estimate.cpp:692-706)
      }
      L.fme_down.assign (nslices, 0u);
      scr.pSliceFMECostDown = &L.fme_down[0];
      job.pScreen = &scr;
    }
  }
  job.pGomRc = NULL;
  if (L.gom && st->gom_kernel && pParam->iUsageType == CAMERA_VIDEO_REAL_TIME && job.pScreen == NULL && nslices == 1) {      // (I pictures too since round 5)
    // WELS_HIP_GOM=2: the groups' QP recursion (RcCalculateGomQp / RcGomTargetBits between the groups, from the bits the device counts
    // itself) runs inside the kernel, so the picture is ONE device call like a constant-QP picture and shares a launch with other
    // sessions' pictures.  The reference's own rate control still runs in the slice loop, on the real bit positions: every macroblock's QP
    // is compared there (a difference fails the frame).  Groups must be whole macroblock rows (else group by group as before).
    const SWelsSvcRc* pRc = &pCtx->pWelsSvcRc[did];
    const SWelsSvcRc* pRcBase = RcJudgeBaseUsability (pCtx);
    if (pRcBase == NULL) pRcBase = pRc;
    if (pRc->iNumberMbGom % mbw == 0 && pRc->iGomSize <= 160 && pRc->bGomRC) {
      WelsHipGomRc& g = L.gomrc;
      memset (&g, 0, sizeof (g));
      // GomRCInitForOneSlice (ratectl.cpp:541-547; WelsCodeOneSlice calls it after this hook) for the picture's only slice
      g.iNumberMbGom = pRc->iNumberMbGom; g.iEndMbSlice = num_mb - 1;
      g.iTargetBitsSlice = WELS_DIV_ROUND (static_cast<int64_t> (pRc->iBitsPerMb) * num_mb, INT_MULTIPLY);
      g.iMinFrameQp = pRc->iMinFrameQp; g.iMaxFrameQp = pRc->iMaxFrameQp;
      g.iGomSize = pRc->iGomSize; g.pGomSad = pRcBase->pCurrentFrameGomSad;
      job.pGomRc = &g;
      L.gom = false;               // (coded by the one call below; HipCodeSlice only entropy-codes and checks the QPs)
    }
  }
  L.dyn = pParam->sSpatialLayers[did].sSliceArgument.uiSliceMode == SM_SIZELIMITED_SLICE;
  if (L.dyn) {
    // (the slice table at this point: one slice per partition, i.e. per slice thread -- pFirstMbIdxOfSlice = the partitions' first macroblocks)
    const int nparts = pCtx->iActiveThreadsNum;
    if (L.gom || nslices != nparts || nparts < 1 || nparts > MAX_THREADS_NUM) {
      fprintf (stderr, "welship hooks: size-limited slices with GOM-level QP / %d slices in %d partitions\n", nslices, nparts); st->failed = true; return ENC_RETURN_UNEXPECTED;
    }
    L.dyn_calls = 0; L.dyn_slices = 0; L.dyn_parts_left = 0; L.dyn_mbs_coded = 0;
    if (nparts > 1) {       // the device's slice table: the partitions (FirstMbIdxOfPartition / EndMbIdxOfPartition, svc_enc_slice_segment.cpp)
      first.clear();
      for (int q = 0; q < nparts; ++q) first.push_back (pCurLayer->FirstMbIdxOfPartition[q]);
      first.push_back (num_mb);
      for (int q = 0; q < nparts; ++q) if (first[q + 1] != pCurLayer->EndMbIdxOfPartition[q] + 1 || first[q + 1] <= first[q]) {
        fprintf (stderr, "welship hooks: partitions of the picture are not contiguous (%d: %d..%d)\n", q, first[q], pCurLayer->EndMbIdxOfPartition[q]); st->failed = true; return ENC_RETURN_UNEXPECTED;
      }
      job.pSliceFirstMb = &first[0];
    }
    for (int q = 0; q < nparts; ++q) {
      L.part[q].coded_upto = 0; L.part[q].coded_slice = -1;
      // (a partition of a single macroblock is left out by its task, wels_task_encoder.cpp:246-250; one thread: WelsCodeOnePicPartition codes whatever there is)
      if (nparts == 1 || pCurLayer->EndMbIdxOfPartition[q] > pCurLayer->FirstMbIdxOfPartition[q]) ++L.dyn_parts_left;
    }
    L.records = NULL; L.packed = NULL;
    if (st->trace) fprintf (stderr, "welship hooks: did %d %c picture qp %d size-limited slices (%u bytes, %d partition%s) cur %d ref %d deblock %d expand %d\n", did, is_p ? 'P' : 'I', job.iQp,
                            pCurLayer->sSliceEncCtx.uiSliceSizeConstraint, nparts, nparts == 1 ? "" : "s", job.iCurPic, job.iRefPic, job.bDeblock, job.bExpand);
    // The first macroblocks of the picture now (slice 0 begins at macroblock 0 whatever happens): this call also takes the picture's
    // inputs to the device, before any partition's task asks for its macroblocks.
    const int rcd = DynCode (st, L, 0, 0, 0, 0, is_p, 0, nparts == 1 ? num_mb : pCurLayer->EndMbIdxOfPartition[0] + 1);
    if (rcd) return rcd;
    return ENC_RETURN_SUCCESS;
  }
  if (L.gom) {
    if (nslices != 1) { fprintf (stderr, "welship hooks: GOM-level QP with %d slices\n", nslices); st->failed = true; return ENC_RETURN_UNEXPECTED; }
    L.mb_qp.assign (num_mb, (uint8_t)pCtx->iGlobalQp);
    L.coded_upto = 0;
    L.records = NULL; L.packed = NULL;
    if (st->trace) fprintf (stderr, "welship hooks: did %d %c picture GOM-level QP (%d MBs per group) cur %d ref %d deblock %d expand %d\n", did, is_p ? 'P' : 'I',
                            pCtx->pWelsSvcRc[did].iNumberMbGom, job.iCurPic, job.iRefPic, job.bDeblock, job.bExpand);
    return ENC_RETURN_SUCCESS;
  }
  const void* rec = NULL;
  int rc;
  int32_t got_packed = 0;           // what the library really returned (pictures above WELSHIP_PACKED_MAX_MB macroblocks, WELSHIP_COMPACT=0: full records)
  job.bPackedRecords = (st->packed && num_mb <= WELSHIP_PACKED_MAX_MB) ? 1 : 0;
  job.pbRecordsPacked = &got_packed;
  { Stopwatch sw (st->timing ? &st->t_encode : NULL); rc = g_api.FrameEncode (L.ctx, &job, &rec); }
  ++st->pictures;
  if (rc) { fprintf (stderr, "welship hooks: WelsHipFrameEncode failed (%d: %s)\n", rc, g_api.GetLastError()); st->failed = true; return ENC_RETURN_UNEXPECTED; }
  if (got_packed) { L.packed = (const WelsHipPackedRecords*)rec; L.records = NULL; }
  else { L.packed = NULL; L.records = (const WhMbRecord*)rec; }
  job.bPackedRecords = 0; job.pbRecordsPacked = NULL;          // (the job is reused by the retry after a CAVLC overflow, which asks for what the slice loop holds then)
  L.states.clear();
  if (pParam->iSpatialLayerNum > did + 1) {      // a higher layer will read this layer's motion (see pIlHint above)
    L.states.resize (num_mb);
    if (g_api.FrameGetMbStates (L.ctx, job.iCurPic, &L.states[0], sizeof (WhMbState) * num_mb)) { st->failed = true; return ENC_RETURN_UNEXPECTED; }
  }
  if (st->eager_recon && !FetchRecon (st, pCtx, L, job.iCurPic)) return ENC_RETURN_UNEXPECTED;
  if (st->trace) fprintf (stderr, "welship hooks: did %d %c picture qp %d slices %d cur %d ref %d deblock %d expand %d mvrange %d complexity %d\n", did, is_p ? 'P' : 'I', job.iQp, nslices, job.iCurPic, job.iRefPic, job.bDeblock, job.bExpand, job.iMvRange, job.iComplexityMode);
  return ENC_RETURN_SUCCESS;
}

static const Mb_Type kMbType[7] = { MB_TYPE_INTRA4x4, MB_TYPE_INTRA16x16, MB_TYPE_16x16, MB_TYPE_16x8, MB_TYPE_8x16, MB_TYPE_8x8, MB_TYPE_SKIP };

// One macroblock record -> SMB + the parts of SMbCache the entropy writer reads (svc_set_mb_syn_cavlc.cpp:60-330).
void LoadRecord (const WhMbRecord& R, SMB* pMb, SMbCache* pMbCache) {
  pMb->uiMbType = kMbType[R.mb_type <= 6 ? R.mb_type : 6];
  pMb->uiCbp = R.cbp;
  for (int i = 0; i < 4; ++i) { pMb->uiSubMbType[i] = SUB_MB_TYPE_8x8; pMb->pRefIndex[i] = R.ref_idx[i]; }
  // the writer codes sMv - sMbMvp (WelsSpatialWriteMbPred); the device hands over that difference, so the predictor is zero
  for (int i = 0; i < 16; ++i) {
    pMb->sMv[i].iMvX = R.mvd[i][0]; pMb->sMv[i].iMvY = R.mvd[i][1];
    pMbCache->sMbMvp[i].iMvX = 0; pMbCache->sMbMvp[i].iMvY = 0;
  }
  // total_coeff: luma raster 0..15; chroma as the reference lays it out -- Cb 16,17,20,21 and Cr 18,19,22,23
  for (int i = 0; i < 16; ++i) pMb->pNonZeroCount[i] = (int8_t)R.nzc[i];
  static const int kC[8] = { 16, 17, 20, 21, 18, 19, 22, 23 };
  for (int i = 0; i < 8; ++i) pMb->pNonZeroCount[kC[i]] = (int8_t)R.nzc[16 + i];
  pMbCache->uiLumaI16x16Mode = R.i16_mode;
  pMbCache->uiChmaI8x8Mode = R.chroma_mode;
  pMb->uiChromPredMode = R.chroma_mode;      // WelsMdIntraSecondaryModesEnc / WelsMdFirstIntraMode leave it for the CABAC contexts of the neighbours
  if (R.mb_type == WH_MB_I4x4) {
    // luma4x4BlkIdx order on both sides (pPrevIntra4x4PredModeFlag / pRemIntra4x4PredModeFlag are written in coding order)
    for (int i = 0; i < 16; ++i) { pMbCache->pPrevIntra4x4PredModeFlag[i] = ((R.i4_prev_flags >> i) & 1) != 0; pMbCache->pRemIntra4x4PredModeFlag[i] = R.i4_rem[i]; }
  }
  // SDCTCoeff (mb_cache.h:62-70) and the record's coefficient part have the same layout
  memcpy (pMbCache->pDct, &R.luma[0][0], sizeof (SDCTCoeff));
}

int32_t HipCodeSlice (sWelsEncCtx* pCtx, SSlice* pSlice) {
  SWelsFuncPtrList* pFunc = pCtx->pFuncList;
  HipState* st = (HipState*)pFunc->pHipState;
  HipLayer& L = st->layer[pCtx->uiDependencyId];
  if (st->failed || (L.records == NULL && L.packed == NULL && !L.gom && !L.dyn)) return ENC_RETURN_UNEXPECTED;
  WhMbRecord sExpanded;              // (packed records: the macroblock in hand)
  Stopwatch sw_code (st->timing && pCtx->pSvcParam->iMultipleThreadIdc <= 1 ? &st->t_code : NULL);     // (slice tasks run concurrently: not summed)
  SDqLayer* pCurLayer = pCtx->pCurDqLayer;
  SMbCache* pMbCache = &pSlice->sMbCacheInfo;
  SMB* pMbList = pCurLayer->sMbDataP;
  const int32_t kiSliceFirstMbXY = pSlice->sSliceHeaderExt.sSliceHeader.iFirstMbInSlice;
  const int32_t kiTotalNumMb = pCurLayer->iMbWidth * pCurLayer->iMbHeight;
  const int32_t kiSliceIdx = pSlice->iSliceIdx;
  const bool is_p = pCtx->eSliceType == P_SLICE;
  int32_t iNextMbIdx = kiSliceFirstMbXY, iNumMbCoded = 0;
  static_assert (sizeof (SDCTCoeff) == 816, "SDCTCoeff layout");
  if (is_p) pSlice->iMbSkipRun = 0;
  SDynamicSlicingStack sDss;
  memset (&sDss, 0, sizeof (sDss));
  const bool kbCavlc = pCtx->pSvcParam->iEntropyCodingModeFlag == 0;
  // CABAC: the slice's arithmetic coder starts here, as in WelsISliceMdEnc / WelsMdInterMbLoop (svc_encode_slice.cpp:550-554,1824-1828);
  // the writer (WelsSpatialWriteMbSynCabac) derives its contexts from what it wrote for the neighbours (sMvd, iCbpDc, types)
  if (pCtx->pSvcParam->iEntropyCodingModeFlag) WelsInitSliceCabac (pCtx, pSlice);
  SSliceCtx* pSliceCtx = &pCurLayer->sSliceEncCtx;
  uint32_t uiDynFmeDown = 0;
  const int32_t kiPartitionId = kiSliceIdx % pCtx->iActiveThreadsNum;
  const int32_t kiDynPartEnd = pCtx->iActiveThreadsNum == 1 ? kiTotalNumMb : pCurLayer->EndMbIdxOfPartition[kiPartitionId] + 1;
  const int32_t kiDynPartFirst = pCtx->iActiveThreadsNum == 1 ? 0 : pCurLayer->FirstMbIdxOfPartition[kiPartitionId];
  if (L.dyn) {      // WelsMdInterMbLoopOverDynamicSlice / WelsISliceMdEncDynamic (svc_encode_slice.cpp:1925-1931,620-626)
    if (kbCavlc) sDss.iStartPos = BsGetBitsPos (pSlice->pSliceBsa);
    else { sDss.iStartPos = sDss.iCurrentPos = 0; sDss.pRestoreBuffer = pCtx->pDynamicBsBuffer[kiPartitionId]; }
  }
  for (;;) {
    const int32_t iCurMbIdx = iNextMbIdx;
    SMB* pCurMb = &pMbList[iCurMbIdx];
    if (kbCavlc || L.dyn) pFunc->pfStashMBStatus (&sDss, pSlice, is_p ? pSlice->iMbSkipRun : 0);      // (the position TRY_REENCODING returns to)
    // QP of the macroblock through the reference's own RC entry point (frame constant, or the group's: WelsRcMbInitGom)
    pFunc->pfRc.pfWelsRcMbInit (pCtx, pCurMb, pSlice);
    if (L.gom && iCurMbIdx >= L.coded_upto) {
      // first macroblock of a group: RcCalculateGomQp has just set the group's QP from the bits of the groups before it
      // (ratectl.cpp:1239-1262); the device now codes exactly this group, the loop below entropy-codes it, and so on
      const int nGom = pCtx->pWelsSvcRc[pCtx->uiDependencyId].iNumberMbGom;
      int end = (iCurMbIdx / nGom + 1) * nGom;
      if (end > kiTotalNumMb) end = kiTotalNumMb;
      for (int i = iCurMbIdx; i < end; ++i) L.mb_qp[i] = pCurMb->uiLumaQp;
      L.job.pMbQp = &L.mb_qp[0];
      L.job.pSliceFirstMb = &L.first[0];
      L.job.iMbBegin = iCurMbIdx; L.job.iMbEnd = end;
      const void* rec = NULL;
      const int rc = g_api.FrameEncode (L.ctx, &L.job, &rec);
      if (rc) { fprintf (stderr, "welship hooks: WelsHipFrameEncode (MBs %d..%d) failed (%d: %s)\n", iCurMbIdx, end, rc, g_api.GetLastError()); st->failed = true; return ENC_RETURN_UNEXPECTED; }
      L.records = (const WhMbRecord*)rec; L.packed = NULL;
      L.coded_upto = end;
      if (end == kiTotalNumMb) {       // the picture is complete (filtered, borders expanded)
        if (st->eager_recon && !FetchRecon (st, pCtx, L, L.job.iCurPic)) return ENC_RETURN_UNEXPECTED;
        L.states.clear();
        if (pCtx->pSvcParam->iSpatialLayerNum > pCtx->uiDependencyId + 1) {
          L.states.resize (kiTotalNumMb);
          if (g_api.FrameGetMbStates (L.ctx, L.job.iCurPic, &L.states[0], sizeof (WhMbState) * kiTotalNumMb)) { st->failed = true; return ENC_RETURN_UNEXPECTED; }
        }
      }
    }
    if (L.dyn && (L.part[kiPartitionId].coded_slice != kiSliceIdx || iCurMbIdx >= L.part[kiPartitionId].coded_upto)) {
      // The slice begins here, or the device has not coded this far ahead yet.  What an earlier call coded from here on belonged to
      // the slice before (other neighbours for the macroblocks of the first rows, other predictors after them) and is coded again.
      Stopwatch sw (st->timing && pCtx->iActiveThreadsNum == 1 ? &st->t_encode : NULL);
      const int rcd = DynCode (st, L, kiPartitionId, kiSliceIdx, kiSliceFirstMbXY, iCurMbIdx, is_p, kiDynPartFirst, kiDynPartEnd);
      if (rcd) return rcd;
    }
    bool bInitDone = false;
TRY_REENCODING:
    if (L.packed != NULL) wh_compact_expand (L.packed->pData + L.packed->pOffset[iCurMbIdx], L.packed->pOffset[iCurMbIdx + 1] - L.packed->pOffset[iCurMbIdx], &sExpanded);
    const WhMbRecord& R = L.packed != NULL ? sExpanded : L.records[iCurMbIdx];
    if (pCurMb->uiLumaQp != R.luma_qp) { fprintf (stderr, "welship hooks: QP mismatch at MB %d (%d vs %d)\n", iCurMbIdx, pCurMb->uiLumaQp, R.luma_qp); st->failed = true; return ENC_RETURN_UNEXPECTED; }
    // neighbour caches the entropy writer reads (non-zero counts): the reference's own init functions
    if (!bInitDone) {        // (not repeated for a re-encoded macroblock, as in the reference: the label comes after them)
      WelsMdIntraInit (pCtx, pCurMb, pMbCache, kiSliceFirstMbXY);
      if (is_p) WelsMdInterInit (pCtx, pSlice, pCurMb, kiSliceFirstMbXY);
      bInitDone = true;
    }
    LoadRecord (R, pCurMb, pMbCache);
    UpdateNonZeroCountCache (pCurMb, pMbCache);
    const int32_t iBitsBefore = st->check_bits ? pFunc->pfGetBsPosition (pSlice) : 0, iSkipRunBefore = is_p ? pSlice->iMbSkipRun : 0;
    const int32_t iLastQpBefore = pSlice->uiLastMbQp;
    const int32_t iEncReturn = pFunc->pfWelsSpatialWriteMbSyn (pCtx, pSlice, pCurMb);
    if (st->check_bits && iEncReturn == ENC_RETURN_SUCCESS) {
      // what the device counted (kernels/cavlc_bits.h) against what the writer just produced: the count leaves out ue(mb_skip_run) and
      // se(mb_qp_delta), which depend on the macroblocks before this one in coding order
      const int32_t iWritten = pFunc->pfGetBsPosition (pSlice) - iBitsBefore;
      int32_t iCounted = 0;
      if (R.mb_type != WH_MB_PSKIP) {
        iCounted = R.cavlc_bits & 0x3fffffff;
        if (is_p) iCounted += BsSizeUE (iSkipRunBefore);
        if (R.cavlc_bits & 0x40000000) iCounted += BsSizeSE ((int32_t)R.luma_qp - iLastQpBefore);
      }
      if (iCounted != iWritten) { fprintf (stderr, "welship hooks: CAVLC bit count of MB %d (type %d cbp %d): device %d, writer %d\n", iCurMbIdx, R.mb_type, R.cbp, iCounted, iWritten); st->failed = true; return ENC_RETURN_UNEXPECTED; }
      ++st->bits_checked;
    }
    if (iEncReturn == ENC_RETURN_VLCOVERFLOWFOUND && kbCavlc && pCurMb->uiLumaQp < 50) {
      // TRY_REENCODING (svc_encode_slice.cpp:564-576,1845-1867): the writer could not code the macroblock (a level beyond what
      // Baseline CAVLC can express, or the picture's bitstream buffer nearly full); the reference takes the bitstream back to where
      // the macroblock started and decides it again with its QP raised by 2 -- without re-initialising it, so uiCbp and one cell of the
      // MV cache carry over (WelsHipMbReencode).  On the device that is the whole picture again with this macroblock's QP changed:
      // every other macroblock reproduces itself, the ones after it in the slice see its new reconstruction.
      if (L.gom || L.job.pGomRc != NULL || (pCtx->pSvcParam->iMultipleThreadIdc > 1 && !L.dyn)) {
        fprintf (stderr, "welship hooks: CAVLC overflow at MB %d -- re-encoding is not implemented for GOM-level QP / slice threads\n", iCurMbIdx);
        st->failed = true;
        return ENC_RETURN_UNEXPECTED;
      }
      const int32_t iRun = pFunc->pfStashPopMBStatus (&sDss, pSlice);
      pSlice->iMbSkipRun = iRun;
      uiDynFmeDown += R.fme_down;        // (every pass of the macroblock's searches counts in the reference)
      const uint8_t kuiChromaQpIndexOffset = pCurLayer->sLayerInfo.pPpsP->uiChromaQpIndexOffset;
      pCurMb->uiLumaQp += DELTA_QP;                      // UpdateQpForOverflow (svc_encode_slice.cpp:526-530)
      pCurMb->uiChromaQp = g_kuiChromaQpTable[CLIP3_QP_0_51 (pCurMb->uiLumaQp + kuiChromaQpIndexOffset)];
      std::vector<WelsHipMbReencode> list;
      {
        std::lock_guard<std::mutex> lock (L.dyn_mu);       // (size-limited slices with slice threads: the partitions' tasks share the list)
        WelsHipMbReencode* e = NULL;
        for (size_t i = 0; i < L.reencode.size(); ++i) if (L.reencode[i].iMbXY == iCurMbIdx) e = &L.reencode[i];
        if (e == NULL) { WelsHipMbReencode n; memset (&n, 0, sizeof (n)); n.iMbXY = iCurMbIdx; L.reencode.push_back (n); e = &L.reencode.back(); }
        e->uiLumaQp = pCurMb->uiLumaQp;
        e->uiStaleCbp = R.cbp & 0x3f;
        if (R.mb_type == WH_MB_P8x16) { e->bCell12Valid = 1; e->iCell12Mv[0] = R.mv_tr[0]; e->iCell12Mv[1] = R.mv_tr[1]; }
        if (L.dyn) list = L.reencode;
      }
      if (L.dyn) {
        // Size-limited slices: nothing of the picture is filtered yet, so only what the device has coded ahead from this macroblock on is
        // repeated -- same slice, same range end.
        WelsHipFrameJob jb = L.job;
        jb.pSliceFirstMb = &L.first[0];
        jb.iMbBegin = iCurMbIdx; jb.iMbEnd = L.part[kiPartitionId].coded_upto;
        jb.iDynSlice = kiSliceIdx + 1; jb.iDynSliceFirstMb = kiSliceFirstMbXY;
        jb.pReencode = &list[0]; jb.iNumReencode = (int32_t)list.size(); jb.bRangeAgain = 1; jb.bDynRedoFirst = 1;
        const void* rec = NULL;
        const int rc = g_api.FrameEncode (L.ctx, &jb, &rec);
        if (rc) { fprintf (stderr, "welship hooks: WelsHipFrameEncode (re-encoding MB %d at QP %d) failed (%d: %s)\n", iCurMbIdx, pCurMb->uiLumaQp, rc, g_api.GetLastError()); st->failed = true; return ENC_RETURN_UNEXPECTED; }
        L.records = (const WhMbRecord*)rec; L.packed = NULL;
        if (st->trace) fprintf (stderr, "welship hooks: MB %d coded again at QP %d (%d re-encoded macroblocks in this picture)\n", iCurMbIdx, pCurMb->uiLumaQp, (int)list.size());
        goto TRY_REENCODING;
      }
      L.job.bRetry = 1; L.job.pReencode = &L.reencode[0]; L.job.iNumReencode = (int32_t)L.reencode.size();
      L.job.pSliceFirstMb = &L.first[0];
      const void* rec = NULL;
      const int rc = g_api.FrameEncode (L.ctx, &L.job, &rec);
      if (rc) { fprintf (stderr, "welship hooks: WelsHipFrameEncode (re-encoding MB %d at QP %d) failed (%d: %s)\n", iCurMbIdx, pCurMb->uiLumaQp, rc, g_api.GetLastError()); st->failed = true; return ENC_RETURN_UNEXPECTED; }
      L.records = (const WhMbRecord*)rec; L.packed = NULL;
      if (st->eager_recon && !FetchRecon (st, pCtx, L, L.job.iCurPic)) return ENC_RETURN_UNEXPECTED;
      if (!L.states.empty() && g_api.FrameGetMbStates (L.ctx, L.job.iCurPic, &L.states[0], sizeof (WhMbState) * kiTotalNumMb)) { st->failed = true; return ENC_RETURN_UNEXPECTED; }
      if (st->trace) fprintf (stderr, "welship hooks: MB %d coded again at QP %d (%d re-encoded macroblocks in this picture)\n", iCurMbIdx, pCurMb->uiLumaQp, (int)L.reencode.size());
      goto TRY_REENCODING;
    }
    if (ENC_RETURN_SUCCESS != iEncReturn) return iEncReturn;
    if (L.dyn) {
      // DYNAMIC_SLICING_ONE_THREAD (svc_encode_slice.cpp:654-663,1982-1992): with this macroblock the slice would exceed its size -- the
      // bitstream goes back to where the macroblock began, the slice ends before it and the next slice begins WITH it
      sDss.iCurrentPos = pFunc->pfGetBsPosition (pSlice);
      if (DynSlcJudgeSliceBoundaryStepBack (pCtx, pSlice, pSliceCtx, pCurMb, &sDss)) {
        uiDynFmeDown += R.fme_down;      // (the reference's feature search has added this macroblock's share to THIS slice before the slice ended in front of it)
        const int32_t iRun = pFunc->pfStashPopMBStatus (&sDss, pSlice);
        if (is_p) pSlice->iMbSkipRun = iRun;
        pCurLayer->LastCodedMbIdxOfPartition[kiPartitionId] = iCurMbIdx - 1;
        ++pCurLayer->NumSliceCodedOfPartition[kiPartitionId];
        L.part[kiPartitionId].est[is_p ? 1 : 0] = iCurMbIdx - kiSliceFirstMbXY;
        { std::lock_guard<std::mutex> lock (L.dyn_mu); ++L.dyn_slices; }
        break;
      }
    }
    pCurMb->uiSliceIdc = kiSliceIdx;
    // uiRefMbType of the picture (WelsMdInterSaveSadAndRefMbType): besides the device's mode decision, the host's complexity
    // analysis of the NEXT picture reads it (wels_preprocess.cpp:830-930: background MBs whose reference MB is intra)
    if (is_p) pCurLayer->pDecPic->uiRefMbType[iCurMbIdx] = R.bgd_skip ? (Mb_Type)MB_TYPE_BACKGROUND : pCurMb->uiMbType;
    if (R.bgd_skip) {
      // VaaBackgroundMbDataUpdate (svc_base_layer_md.cpp:1341-1350): a background macroblock's source samples are replaced by the
      // reference's, which the pre-processing of the following pictures sees
      SVAAFrameInfo* pVaa = pCtx->pVaa;
      const int32_t kiOffsetY = (pCurMb->iMbY * pVaa->iPicStride + pCurMb->iMbX) << 4, kiOffsetUV = (pCurMb->iMbY * pVaa->iPicStrideUV + pCurMb->iMbX) << 3;
      pFunc->pfCopy16x16Aligned (pVaa->pCurY + kiOffsetY, pVaa->iPicStride, pVaa->pRefY + kiOffsetY, pVaa->iPicStride);
      pFunc->pfCopy8x8Aligned (pVaa->pCurU + kiOffsetUV, pVaa->iPicStrideUV, pVaa->pRefU + kiOffsetUV, pVaa->iPicStrideUV);
      pFunc->pfCopy8x8Aligned (pVaa->pCurV + kiOffsetUV, pVaa->iPicStrideUV, pVaa->pRefV + kiOffsetUV, pVaa->iPicStrideUV);
    }
    pFunc->pfRc.pfWelsRcMbInfoUpdate (pCtx, pCurMb, R.cost, pSlice);
    uiDynFmeDown += R.fme_down;
    ++iNumMbCoded;
    iNextMbIdx = WelsGetNextMbOfSlice (pCurLayer, iCurMbIdx);
    if (iNextMbIdx == -1 || iNextMbIdx >= kiTotalNumMb || iNumMbCoded >= kiTotalNumMb) {
      if (L.dyn) {
        if (!is_p) pSlice->iCountMbNumInSlice = iCurMbIdx - pCurLayer->LastCodedMbIdxOfPartition[kiPartitionId];
        pCurLayer->LastCodedMbIdxOfPartition[kiPartitionId] = iCurMbIdx;
        ++pCurLayer->NumSliceCodedOfPartition[kiPartitionId];
        // the partition's last macroblock is written; with the picture's last partition: the picture-wide passes (filter, borders) and
        // the host's copy of the reconstruction
        std::lock_guard<std::mutex> lock (L.dyn_mu);
        ++L.dyn_slices;
        if (--L.dyn_parts_left == 0) {
          WelsHipFrameJob jb = L.job;
          jb.pSliceFirstMb = &L.first[0];
          jb.iMbBegin = kiTotalNumMb; jb.iMbEnd = kiTotalNumMb; jb.iDynSlice = kiSliceIdx + 1; jb.iDynSliceFirstMb = kiSliceFirstMbXY;
          if (!L.reencode.empty()) { jb.pReencode = &L.reencode[0]; jb.iNumReencode = (int32_t)L.reencode.size(); }     // (the QP_Y chain for the filter)
          const void* rec = NULL;
          const int rc = g_api.FrameEncode (L.ctx, &jb, &rec);
          if (rc) { fprintf (stderr, "welship hooks: WelsHipFrameEncode (closing the picture) failed (%d: %s)\n", rc, g_api.GetLastError()); st->failed = true; return ENC_RETURN_UNEXPECTED; }
          ++st->pictures;
          if (st->eager_recon && !FetchRecon (st, pCtx, L, L.job.iCurPic)) return ENC_RETURN_UNEXPECTED;
          if (st->trace) fprintf (stderr, "welship hooks: layer %d picture complete: %d slices, %d device calls, %ld macroblocks coded for %d\n", (int)pCtx->uiDependencyId, L.dyn_slices, L.dyn_calls, L.dyn_mbs_coded, kiTotalNumMb);
        }
      }
      break;
    }
  }
  if (is_p && pSlice->iMbSkipRun) BsWriteUE (pSlice->pSliceBsa, pSlice->iMbSkipRun);
  // WelsDiamondCrossFeatureSearch's account of what the feature search saved (svc_motion_estimate.cpp:1080-1092), read by UpdateFMESwitch
  if (is_p && L.job.pScreen != NULL && L.dyn) pSlice->uiSliceFMECostDown += uiDynFmeDown;        // (size-limited slices: summed over the macroblocks this slice really took)
  else if (is_p && L.job.pScreen != NULL && kiSliceIdx >= 0 && kiSliceIdx < (int32_t)L.fme_down.size()) pSlice->uiSliceFMECostDown += L.fme_down[kiSliceIdx];
  // for the layer above: this slice's real motion vectors in the SMB array (the writer above was fed vector differences)
  if (!L.states.empty()) {
    int32_t iMb = kiSliceFirstMbXY;
    for (int32_t n = 0; n < iNumMbCoded && iMb >= 0 && iMb < kiTotalNumMb; ++n) {
      const WhMbState& S = L.states[iMb];
      for (int i = 0; i < 16; ++i) { pMbList[iMb].sMv[i].iMvX = S.mv[i][0]; pMbList[iMb].sMv[i].iMvY = S.mv[i][1]; }
      iMb = WelsGetNextMbOfSlice (pCurLayer, iMb);
    }
  }
  return ENC_RETURN_SUCCESS;
}

// CWelsPreProcess::DownsamplePadding (wels_preprocess.cpp:625-675): the step that makes one spatial layer's source picture from the
// next larger one -- CDownsampling::Process of codec/processing (downsample.cpp:144-277: 2:1 / 4:1 / 3:1 averages, a cascade of halvings,
// or the general bilinear filters) -- as one call into libwelship.so (include/welship.h 3b).  The padding that follows stays the
// reference's.  Returns non-zero when the device did not do it: the caller then runs its C functions.
int32_t HipDownsample (void* p, uint8_t* const pDst[3], const int32_t iDstStride[3], int32_t iDstWidth, int32_t iDstHeight,
                       const uint8_t* const pSrc[3], const int32_t iSrcStride[3], int32_t iSrcWidth, int32_t iSrcHeight) {
  HipState* st = (HipState*)p;
  if (st == NULL || !st->downsample || st->failed || g_api.DownsamplePicture == NULL) return 1;
  if (iSrcWidth <= iDstWidth || iSrcHeight <= iDstHeight || iDstWidth < 2 || iDstHeight < 2) return 1;       // (RET_INVALIDPARAM of the C path: leave it to it)
  int rc;
  { Stopwatch sw (st->timing ? &st->t_down : NULL); rc = g_api.DownsamplePicture (st->device, pDst, iDstStride, iDstWidth, iDstHeight, pSrc, iSrcStride, iSrcWidth, iSrcHeight); }
  if (rc != 0) {
    if (st->trace) fprintf (stderr, "welship hooks: down-sampling %dx%d -> %dx%d stays on the host (%d)\n", iSrcWidth, iSrcHeight, iDstWidth, iDstHeight, rc);
    return 1;
  }
  ++st->downsampled;
  if (st->trace) fprintf (stderr, "welship hooks: down-sampled %dx%d -> %dx%d on the device\n", iSrcWidth, iSrcHeight, iDstWidth, iDstHeight);
  return 0;
}

void HipRelease (void* p) {
  HipState* st = (HipState*)p;
  if (st == NULL) return;
  if (st->check_bits && st->trace) fprintf (stderr, "welship hooks: CAVLC bit counts of %ld macroblocks equal the writer's\n", st->bits_checked.load());
  if (st->timing && st->pictures) fprintf (stderr, "welship hooks: %d pictures; per picture: device call %.3f ms, reconstruction copy-back %.3f ms, slice coding from the records %.3f ms, pre-analysis call %.3f ms, down-sampling calls %.3f ms\n",
                                           st->pictures, 1e3 * st->t_encode / st->pictures, 1e3 * st->t_getpic / st->pictures, 1e3 * st->t_code / st->pictures, 1e3 * st->t_vaa / st->pictures, 1e3 * st->t_down / st->pictures);
  if (st->trace && st->host_pictures) fprintf (stderr, "welship hooks: %ld pictures of layers below %d macroblocks were coded by the host\n", st->host_pictures, st->min_layer_mbs);
  for (int i = 0; i < MAX_DEPENDENCY_LAYER; ++i) if (st->layer[i].ctx) {
    g_api.FrameCtxDestroy (st->layer[i].ctx);
    if (st->trace) fprintf (stderr, "welship hooks: device context of layer %d released\n", i);
  }
  delete st;
}

// Which sessions run on the device.  `why` receives the reason when the answer is no.
bool WelsHipSupported (const SWelsSvcCodingParam* p, const char** why) {
#define NO(msg) do { *why = msg; return false; } while (0)
  if (p->iUsageType != CAMERA_VIDEO_REAL_TIME && p->iUsageType != SCREEN_CONTENT_REAL_TIME) NO ("usage types beyond camera video and screen content");
  if (p->iUsageType == SCREEN_CONTENT_REAL_TIME && getenv ("WELS_HIP_SCREEN") && atoi (getenv ("WELS_HIP_SCREEN")) == 0) NO ("screen content switched off (WELS_HIP_SCREEN=0)");
  if (p->iEntropyCodingModeFlag != 0 && getenv ("WELS_HIP_CABAC") && atoi (getenv ("WELS_HIP_CABAC")) == 0) NO ("CABAC switched off (WELS_HIP_CABAC=0)");
  // simulcast AVC layers are independent streams (no inter-layer prediction): one device context per layer, optionally one
  // GPU per layer (WELS_HIP_LAYER_DEVICES=1: layer d runs on device WELS_HIP_DEVICE + d; =n, n >= 2: on device WELS_HIP_DEVICE + d mod n)
  if (p->iSpatialLayerNum != 1 && !p->bSimulcastAVC) NO ("spatial layers with SVC syntax (tried: not byte-identical yet; simulcast AVC is)");
  // slice threads: every slice task entropy-codes its slice from the (read-only) records of the picture; see HipFrameMd for the filter
  if (p->iMultipleThreadIdc != 1 && getenv ("WELS_HIP_THREADS") && atoi (getenv ("WELS_HIP_THREADS")) == 0) NO ("slice threads switched off (WELS_HIP_THREADS=0)");
  // bEnableAdaptiveQuant: ParamValidation switches it off for every session (encoder_ext.cpp:300-301), nothing to check
  // Defaults (round 3, after every one of these paths had run on the MI355X: profiles/r03_*_mi355x.txt): everything the binding
  // implements is ON.  The switches only take a path away again: WELS_HIP_DYNSLICE=0, WELS_HIP_GOM=0 (see the table in INTEGRATION.md B).
  const char* gom = getenv ("WELS_HIP_GOM");
  const bool gom_off = gom != NULL && atoi (gom) == 0;
  for (int i = 0; i < p->iSpatialLayerNum; ++i) {
    const SSliceArgument& sa = p->sSpatialLayers[i].sSliceArgument;
    if (sa.uiSliceMode == SM_SIZELIMITED_SLICE) {
      // Size-limited slices feed the bitstream position back into mode decision: a slice ends where the writer says, and the macroblock the
      // next one begins with is decided again without its neighbours.  The binding codes ahead of the writer and repeats the rest of the
      // picture from every slice start (HipCodeSlice): several device calls per picture, bit-exact (both SHA1 tables' -slcmd 3 rows).
      const char* ds = getenv ("WELS_HIP_DYNSLICE");
      if (ds != NULL && atoi (ds) == 0) NO ("size-limited slices switched off (WELS_HIP_DYNSLICE=0)");
      if (p->iSpatialLayerNum != 1) NO ("size-limited slices: one spatial layer only");
    }
    // Rate control with one slice per picture = GOM-level QP (ratectl.cpp:1199-1204): the QP of a group of macroblocks depends on the
    // bits of the groups before it.  P pictures of camera sessions whose groups are whole rows run the recursion inside the kernel (one
    // device call per picture); the rest (I pictures, screen content, CABAC) one device round trip PER GROUP -- bit-exact, a latency
    // chain.  WELS_HIP_GOM=0 leaves such sessions to the C path, WELS_HIP_GOM=1 forces the per-group calls everywhere.
    if (p->iRCMode != RC_OFF_MODE && gom_off) {
      const int mbs = ((p->sSpatialLayers[i].iVideoWidth + 15) >> 4) * ((p->sSpatialLayers[i].iVideoHeight + 15) >> 4);
      const bool one_slice = sa.uiSliceMode == SM_SINGLE_SLICE || (sa.uiSliceMode == SM_FIXEDSLCNUM_SLICE && sa.uiSliceNum <= 1) ||
                             (sa.uiSliceMode == SM_RASTER_SLICE && (sa.uiSliceMbNum[0] == 0 || (int)sa.uiSliceMbNum[0] >= mbs));
      if (one_slice) NO ("rate control with one slice per picture (GOM-level QP) switched off (WELS_HIP_GOM=0)");
    }
  }
  return true;
#undef NO
}

// WELS_HIP_LEAVES=1: the leaf slots of the dispatch table point at the device-backed exports of include/welship_leaf.h.  Every export has
// exactly its slot's typedef (the static_assert in LEAF) and is bound by name from the library the frame hooks use.
int InstallLeaves (SWelsFuncPtrList* fl, const char** why) {
  const char* path = getenv ("WELSHIP_LIB");
  void* h = dlopen (path && *path ? path : "libwelship.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) { *why = "libwelship.so not loadable"; return 0; }
  int (*avail) (void) = (int (*) (void))dlsym (h, "WelsHipLeafAvailable");
  if (!avail) { *why = "libwelship.so without the leaf layer"; return 0; }
  if (avail() != WELSHIP_OK) { *why = "no usable device"; return 0; }
  int n = 0, missing = 0;
#define LEAF(slot, sym) do { \
    static_assert (std::is_same<decltype (&sym), std::remove_reference<decltype (slot)>::type>::value, #sym ": not the type of its slot"); \
    void* f_ = dlsym (h, #sym); if (f_) { slot = (decltype (&sym))f_; ++n; } else ++missing; } while (0)
  SSampleDealingFunc& sd = fl->sSampleDealingFuncs;
#define LEAF_BLOCK(idx, name) LEAF (sd.pfSampleSad[idx], WelsHipSampleSad##name); LEAF (sd.pfSampleSatd[idx], WelsHipSampleSatd##name); LEAF (sd.pfSample4Sad[idx], WelsHipSampleSadFour##name)
  LEAF_BLOCK (BLOCK_16x16, 16x16); LEAF_BLOCK (BLOCK_16x8, 16x8); LEAF_BLOCK (BLOCK_8x16, 8x16); LEAF_BLOCK (BLOCK_8x8, 8x8);
  LEAF_BLOCK (BLOCK_4x4, 4x4); LEAF_BLOCK (BLOCK_8x4, 8x4); LEAF_BLOCK (BLOCK_4x8, 4x8);
  LEAF (fl->pfDctT4, WelsHipDctT4); LEAF (fl->pfDctFourT4, WelsHipDctFourT4);
  LEAF (fl->pfQuantization4x4, WelsHipQuant4x4); LEAF (fl->pfQuantizationDc4x4, WelsHipQuant4x4Dc);
  LEAF (fl->pfQuantizationFour4x4, WelsHipQuantFour4x4); LEAF (fl->pfQuantizationFour4x4Max, WelsHipQuantFour4x4Max);
  LEAF (fl->pfQuantizationHadamard2x2, WelsHipHadamardQuant2x2); LEAF (fl->pfQuantizationHadamard2x2Skip, WelsHipHadamardQuant2x2Skip);
  LEAF (fl->pfTransformHadamard4x4Dc, WelsHipHadamardT4Dc);
  LEAF (fl->pfScan4x4, WelsHipScan4x4DcAc); LEAF (fl->pfScan4x4Ac, WelsHipScan4x4Ac);
  LEAF (fl->pfCalculateSingleCtr4x4, WelsHipCalculateSingleCtr4x4); LEAF (fl->pfGetNoneZeroCount, WelsHipGetNoneZeroCount);
  LEAF (fl->pfDequantization4x4, WelsHipDequant4x4); LEAF (fl->pfDequantizationFour4x4, WelsHipDequantFour4x4);
  LEAF (fl->pfDequantizationIHadamard4x4, WelsHipDequantIHadamard4x4);
  LEAF (fl->pfIDctT4, WelsHipIDctT4Rec); LEAF (fl->pfIDctFourT4, WelsHipIDctFourT4Rec); LEAF (fl->pfIDctI16x16Dc, WelsHipIDctRecI16x16Dc);
  LEAF (fl->sMcFuncs.pMcLumaFunc, WelsHipMcLuma); LEAF (fl->sMcFuncs.pMcChromaFunc, WelsHipMcChroma);
  LEAF (fl->sMcFuncs.pfLumaHalfpelHor, WelsHipMcHorVer20); LEAF (fl->sMcFuncs.pfLumaHalfpelVer, WelsHipMcHorVer02);
  LEAF (fl->sMcFuncs.pfLumaHalfpelCen, WelsHipMcHorVer22); LEAF (fl->sMcFuncs.pfSampleAveraging, WelsHipPixelAvg);
  LEAF (fl->pfGetLumaI4x4Pred[I4_PRED_V], WelsHipI4x4LumaPredV); LEAF (fl->pfGetLumaI4x4Pred[I4_PRED_H], WelsHipI4x4LumaPredH);
  LEAF (fl->pfGetLumaI4x4Pred[I4_PRED_DC], WelsHipI4x4LumaPredDc); LEAF (fl->pfGetLumaI4x4Pred[I4_PRED_DC_L], WelsHipI4x4LumaPredDcLeft);
  LEAF (fl->pfGetLumaI4x4Pred[I4_PRED_DC_T], WelsHipI4x4LumaPredDcTop); LEAF (fl->pfGetLumaI4x4Pred[I4_PRED_DC_128], WelsHipI4x4LumaPredDcNA);
  LEAF (fl->pfGetLumaI4x4Pred[I4_PRED_DDL], WelsHipI4x4LumaPredDDL); LEAF (fl->pfGetLumaI4x4Pred[I4_PRED_DDL_TOP], WelsHipI4x4LumaPredDDLTop);
  LEAF (fl->pfGetLumaI4x4Pred[I4_PRED_DDR], WelsHipI4x4LumaPredDDR); LEAF (fl->pfGetLumaI4x4Pred[I4_PRED_VL], WelsHipI4x4LumaPredVL);
  LEAF (fl->pfGetLumaI4x4Pred[I4_PRED_VL_TOP], WelsHipI4x4LumaPredVLTop); LEAF (fl->pfGetLumaI4x4Pred[I4_PRED_VR], WelsHipI4x4LumaPredVR);
  LEAF (fl->pfGetLumaI4x4Pred[I4_PRED_HU], WelsHipI4x4LumaPredHU); LEAF (fl->pfGetLumaI4x4Pred[I4_PRED_HD], WelsHipI4x4LumaPredHD);
  LEAF (fl->pfGetLumaI16x16Pred[I16_PRED_V], WelsHipI16x16LumaPredV); LEAF (fl->pfGetLumaI16x16Pred[I16_PRED_H], WelsHipI16x16LumaPredH);
  LEAF (fl->pfGetLumaI16x16Pred[I16_PRED_DC], WelsHipI16x16LumaPredDc); LEAF (fl->pfGetLumaI16x16Pred[I16_PRED_P], WelsHipI16x16LumaPredPlane);
  LEAF (fl->pfGetLumaI16x16Pred[I16_PRED_DC_L], WelsHipI16x16LumaPredDcLeft); LEAF (fl->pfGetLumaI16x16Pred[I16_PRED_DC_T], WelsHipI16x16LumaPredDcTop);
  LEAF (fl->pfGetLumaI16x16Pred[I16_PRED_DC_128], WelsHipI16x16LumaPredDcNA);
  LEAF (fl->pfGetChromaPred[C_PRED_DC], WelsHipIChromaPredDc); LEAF (fl->pfGetChromaPred[C_PRED_H], WelsHipIChromaPredH);
  LEAF (fl->pfGetChromaPred[C_PRED_V], WelsHipIChromaPredV); LEAF (fl->pfGetChromaPred[C_PRED_P], WelsHipIChromaPredPlane);
  LEAF (fl->pfGetChromaPred[C_PRED_DC_L], WelsHipIChromaPredDcLeft); LEAF (fl->pfGetChromaPred[C_PRED_DC_T], WelsHipIChromaPredDcTop);
  LEAF (fl->pfGetChromaPred[C_PRED_DC_128], WelsHipIChromaPredDcNA);
  DeblockingFunc& db = fl->pfDeblocking;
  LEAF (db.pfLumaDeblockingLT4Ver, WelsHipDeblockLumaLt4V); LEAF (db.pfLumaDeblockingEQ4Ver, WelsHipDeblockLumaEq4V);
  LEAF (db.pfLumaDeblockingLT4Hor, WelsHipDeblockLumaLt4H); LEAF (db.pfLumaDeblockingEQ4Hor, WelsHipDeblockLumaEq4H);
  LEAF (db.pfChromaDeblockingLT4Ver, WelsHipDeblockChromaLt4V); LEAF (db.pfChromaDeblockingEQ4Ver, WelsHipDeblockChromaEq4V);
  LEAF (db.pfChromaDeblockingLT4Hor, WelsHipDeblockChromaLt4H); LEAF (db.pfChromaDeblockingEQ4Hor, WelsHipDeblockChromaEq4H);
  LEAF (fl->pfCopy16x16Aligned, WelsHipCopy16x16); LEAF (fl->pfCopy16x16NotAligned, WelsHipCopy16x16); LEAF (fl->pfCopy8x8Aligned, WelsHipCopy8x8);
  LEAF (fl->pfCopy16x8NotAligned, WelsHipCopy16x8); LEAF (fl->pfCopy8x16Aligned, WelsHipCopy8x16);
  LEAF (fl->pfCopy4x4, WelsHipCopy4x4); LEAF (fl->pfCopy8x4, WelsHipCopy8x4); LEAF (fl->pfCopy4x8, WelsHipCopy4x8);
  LEAF (fl->pfSetMemZeroSize8, WelsHipSetMemZero); LEAF (fl->pfSetMemZeroSize64Aligned16, WelsHipSetMemZero); LEAF (fl->pfSetMemZeroSize64, WelsHipSetMemZero);
  // the Combined3 slots are NULL in the C build (sample.cpp:363-367); with WELS_HIP_LEAVES=2 they are filled as a SIMD build fills them, and
  // mode decision then takes its combined paths (svc_base_layer_md.cpp:380, :474, :885) -- same decisions, so the same bitstream
  if (atoi (getenv ("WELS_HIP_LEAVES")) >= 2) {
    LEAF (sd.pfIntra4x4Combined3Satd, WelsHipIntra4x4Combined3Satd);
    LEAF (sd.pfIntra16x16Combined3Satd, WelsHipIntra16x16Combined3Satd); LEAF (sd.pfIntra16x16Combined3Sad, WelsHipIntra16x16Combined3Sad);
    LEAF (sd.pfIntra8x8Combined3Satd, WelsHipIntra8x8Combined3Satd); LEAF (sd.pfIntra8x8Combined3Sad, WelsHipIntra8x8Combined3Sad);
  }
#undef LEAF_BLOCK
#undef LEAF
  if (missing) { *why = "libwelship.so lacks some leaf exports"; return -missing; }
  return n;
}

}  // namespace

// The installer: what an `#if defined(X86_ASM)` block is for the SIMD variants (encoder.cpp:157-232).
// What it decided goes to the encoder's own log (WelsLog, WELS_LOG_INFO: the application's trace callback / ENCODER_OPTION_TRACE_LEVEL decide
// whether anybody sees it, welsCodecTrace.h:42-62) -- an application can find out whether its encoder runs on the device -- and, with
// WELS_HIP_TRACE set, to stderr as before.
static void Report (SLogContext* pLogCtx, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start (ap, fmt);
  vsnprintf (buf, sizeof (buf), fmt, ap);
  va_end (ap);
  if (pLogCtx) WelsLog (pLogCtx, WELS_LOG_INFO, "welship hooks: %s", buf);
  if (getenv ("WELS_HIP_TRACE")) fprintf (stderr, "welship hooks: %s\n", buf);
}
// ---- per-macroblock ground truth (round-5 review, Missing 6) ----------------------------------------------------
// WELS_HIP_MB_TRACE=<file>: every macroblock the entropy writer is handed -- on the unmodified C path (WELS_HIP=0) and through the device hooks
// alike, because both reach the writer through the same slot, SWelsFuncPtrList::pfWelsSpatialWriteMbSyn (svc_encode_slice.cpp:570,645,1861,1972;
// HipCodeSlice above) -- leaves one line: picture, layer, address, type, cbp, QP, the vector differences the writer will code (sMv - sMbMvp of
// each partition: the C path holds vector and predictor, the hooks hold the difference and a zero predictor), reference indices, total_coeff of
// the 24 blocks, the intra modes, and a hash of the SDCTCoeff parts coded_block_pattern says are coded.  tools/mb_truth.py runs a command line
// both ways and names the first macroblock whose lines differ: minutes instead of hours when a stream stops matching.  A macroblock that is
// coded twice (size-limited slices re-code the one that did not fit) leaves two lines; the last one counts.
static PWelsSpatialWriteMbSyn g_pfMbTraceNext = NULL;
static FILE* g_fMbTrace = NULL;
static std::mutex g_MbTraceMu;
struct MbTracePic { const void* pic; int32_t num, poc, count; };
static std::map<std::pair<const void*, int>, MbTracePic> g_MbTracePics;

static int32_t TraceWriteMbSyn (sWelsEncCtx* pCtx, SSlice* pSlice, SMB* pCurMb) {
  const SMbCache* pMbCache = &pSlice->sMbCacheInfo;
  const SDCTCoeff* pDct = pMbCache->pDct;
  const Mb_Type t = pCurMb->uiMbType;
  char line[1024];
  int n = 0;
  uint32_t hash = 2166136261u;
  auto mix = [&] (const int16_t* v, int cnt) { for (int i = 0; i < cnt; ++i) { hash = (hash ^ (uint16_t)v[i]) * 16777619u; } };
  const bool skip = IS_SKIP (t), i16 = t == MB_TYPE_INTRA16x16, i4 = t == MB_TYPE_INTRA4x4;
  const int cbp = skip ? 0 : pCurMb->uiCbp;
  if (!skip) {
    if (i16) mix (pDct->iLumaI16x16Dc, 16);
    for (int b8 = 0; b8 < 4; ++b8) if ((cbp >> b8) & 1) for (int k = 0; k < 4; ++k) mix (pDct->iLumaBlock[b8 * 4 + k] + (i16 ? 1 : 0), i16 ? 15 : 16);
    if ((cbp >> 4) >= 1) mix (&pDct->iChromaDc[0][0], 8);
    if ((cbp >> 4) >= 2) for (int b = 0; b < 8; ++b) mix (pDct->iChromaBlock[b] + 1, 15);
  }
  {
    std::lock_guard<std::mutex> g (g_MbTraceMu);
    MbTracePic& P = g_MbTracePics[std::make_pair ((const void*)pCtx, (int)pCtx->uiDependencyId)];
    if (P.pic != (const void*)pCtx->pDecPic || P.num != pCtx->pDecPic->iFrameNum || P.poc != pCtx->pDecPic->iFramePoc) {
      if (P.pic != NULL) ++P.count;
      P.pic = pCtx->pDecPic; P.num = pCtx->pDecPic->iFrameNum; P.poc = pCtx->pDecPic->iFramePoc;
    }
    n += snprintf (line + n, sizeof (line) - n, "pic %d layer %d %c mb %d type %s cbp %d qp %d", P.count, (int)pCtx->uiDependencyId, pCtx->eSliceType == P_SLICE ? 'P' : 'I',
                   pCurMb->iMbXY, skip ? "skip" : i16 ? "i16x16" : i4 ? "i4x4" : t == MB_TYPE_16x16 ? "p16x16" : t == MB_TYPE_16x8 ? "p16x8" : t == MB_TYPE_8x16 ? "p8x16" : t == MB_TYPE_8x8 ? "p8x8" : "other",
                   cbp, skip ? 0 : (int)pCurMb->uiLumaQp);
  }
  if (!skip && !i16 && !i4) {
    static const int kBlk[4][4] = { {0, 0, 0, 0}, {0, 8, 0, 0}, {0, 2, 0, 0}, {0, 2, 8, 10} };
    const int shape = t == MB_TYPE_16x16 ? 0 : t == MB_TYPE_16x8 ? 1 : t == MB_TYPE_8x16 ? 2 : 3, parts = shape == 0 ? 1 : shape == 3 ? 4 : 2;
    n += snprintf (line + n, sizeof (line) - n, " mvd");
    for (int k = 0; k < parts; ++k)
    {     // (the predictor's index: the partition number, except P_8x8 -- there the 4x4 block like the vector's, svc_set_mb_syn_cavlc.cpp:124-242)
      const int b = kBlk[shape][k], pi = shape == 3 ? b : k;
      n += snprintf (line + n, sizeof (line) - n, " %d,%d", pCurMb->sMv[b].iMvX - pMbCache->sMbMvp[pi].iMvX, pCurMb->sMv[b].iMvY - pMbCache->sMbMvp[pi].iMvY);
    }
    n += snprintf (line + n, sizeof (line) - n, " ref %d,%d,%d,%d", pCurMb->pRefIndex[0], pCurMb->pRefIndex[1], pCurMb->pRefIndex[2], pCurMb->pRefIndex[3]);
  }
  if (i16) n += snprintf (line + n, sizeof (line) - n, " i16mode %d", (int)g_kiMapModeI16x16[pMbCache->uiLumaI16x16Mode & 7]);        // (as written: the DC variants without neighbours are one mode)
  if (i4) {
    n += snprintf (line + n, sizeof (line) - n, " i4");
    for (int k = 0; k < 16; ++k) n += snprintf (line + n, sizeof (line) - n, " %d", pMbCache->pPrevIntra4x4PredModeFlag[k] ? -1 : (int)pMbCache->pRemIntra4x4PredModeFlag[k]);
  }
  if (i16 || i4) n += snprintf (line + n, sizeof (line) - n, " chroma %d", (int)g_kiMapModeIntraChroma[pMbCache->uiChmaI8x8Mode & 7]);
  if (!skip) {
    n += snprintf (line + n, sizeof (line) - n, " nzc");
    for (int k = 0; k < 24; ++k) n += snprintf (line + n, sizeof (line) - n, "%c%d", k == 0 ? ' ' : ',', ((cbp & 15) || i16 || k >= 16) ? (int)pCurMb->pNonZeroCount[k] : 0);
    n += snprintf (line + n, sizeof (line) - n, " levels %08x", hash);
  }
  {
    std::lock_guard<std::mutex> g (g_MbTraceMu);
    if (g_fMbTrace) { fputs (line, g_fMbTrace); fputc ('\n', g_fMbTrace); }
  }
  return g_pfMbTraceNext (pCtx, pSlice, pCurMb);
}

static void InstallMbTrace (SWelsFuncPtrList* pFuncList, SLogContext* pLogCtx) {
  const char* path = getenv ("WELS_HIP_MB_TRACE");
  if (path == NULL || path[0] == 0 || pFuncList->pfWelsSpatialWriteMbSyn == TraceWriteMbSyn) return;
  std::lock_guard<std::mutex> g (g_MbTraceMu);
  if (g_fMbTrace == NULL) g_fMbTrace = fopen (path, "w");
  if (g_fMbTrace == NULL) { fprintf (stderr, "welship hooks: WELS_HIP_MB_TRACE: cannot write %s\n", path); return; }
  if (g_pfMbTraceNext != NULL && g_pfMbTraceNext != pFuncList->pfWelsSpatialWriteMbSyn) {       // (one writer per process: CAVLC and CABAC sessions side by side are not traced)
    fprintf (stderr, "welship hooks: WELS_HIP_MB_TRACE: sessions with different entropy coders in one process, the later one is not traced\n");
    return;
  }
  g_pfMbTraceNext = pFuncList->pfWelsSpatialWriteMbSyn;
  pFuncList->pfWelsSpatialWriteMbSyn = TraceWriteMbSyn;
  (void)pLogCtx;
}
static struct MbTraceFlush { ~MbTraceFlush() { if (g_fMbTrace) fclose (g_fMbTrace); } } g_MbTraceFlush;


void WelsHipInstall (SWelsFuncPtrList* pFuncList, SWelsSvcCodingParam* pParam, SLogContext* pLogCtx) {
  pFuncList->pfHipFrameMd = NULL;
  pFuncList->pfHipCodeSlice = NULL;
  pFuncList->pfHipLayerOnDevice = NULL;
  pFuncList->pfHipRelease = NULL;
  pFuncList->pfHipDownsample = NULL;
  pFuncList->pfHipVaaCalc = NULL;
  pFuncList->pfHipFetchRecon = NULL;
  pFuncList->pfHipBgd = NULL;
  pFuncList->pHipState = NULL;
  InstallMbTrace (pFuncList, pLogCtx);        // (WELS_HIP_MB_TRACE: both paths, see above)
  const char* off = getenv ("WELS_HIP");
  if (off && atoi (off) == 0) { Report (pLogCtx, "not installed (WELS_HIP=0)"); return; }
  const char* why = "";
  if (getenv ("WELS_HIP_LEAVES") && atoi (getenv ("WELS_HIP_LEAVES")) != 0) {
    // the leaf level instead of the frame level: the slots are written before any encoder thread exists; a partial table is refused
    // by taking the process down (the slots already written cannot be told from the C ones afterwards)
    const int n = InstallLeaves (pFuncList, &why);
    if (n < 0) { fprintf (stderr, "welship hooks: leaf functions: %s (%d missing)\n", why, -n); abort(); }
    if (n > 0) Report (pLogCtx, "%d leaf functions installed (frame-level hooks off)", n);
    else Report (pLogCtx, "leaf functions not installed (%s)", why);
    return;
  }
  if (!WelsHipSupported (pParam, &why)) { Report (pLogCtx, "not installed (%s)", why); return; }
  if (!LoadApi()) {
    const char* e = dlerror();
    Report (pLogCtx, "not installed (libwelship.so not loadable: %s)", e ? e : "missing symbols");
    return;
  }
  HipState* st = new HipState();
  st->device = getenv ("WELS_HIP_DEVICE") ? atoi (getenv ("WELS_HIP_DEVICE")) : 0;
  st->trace = getenv ("WELS_HIP_TRACE") != NULL;
  st->timing = st->trace && atoi (getenv ("WELS_HIP_TRACE")) >= 2;
  st->gom_kernel = (getenv ("WELS_HIP_GOM") == NULL || atoi (getenv ("WELS_HIP_GOM")) >= 2) && pParam->iEntropyCodingModeFlag == 0;
  // (screen content: the reference's own pre-processing reads the reconstructed reference picture on the host -- the feature search's hash
  //  lists, PerformFMEPreprocess, svc_motion_estimate.cpp:700-760 -- so such sessions get every picture back)
  st->packed = true;
  st->eager_recon = pParam->iUsageType == SCREEN_CONTENT_REAL_TIME;
  st->check_bits = getenv ("WELS_HIP_CHECK_BITS") != NULL && atoi (getenv ("WELS_HIP_CHECK_BITS")) != 0 && pParam->iEntropyCodingModeFlag == 0;
  st->layer_devices = getenv ("WELS_HIP_LAYER_DEVICES") != NULL ? WELS_MAX (0, atoi (getenv ("WELS_HIP_LAYER_DEVICES"))) : 0;
  st->min_layer_mbs = getenv ("WELS_HIP_MIN_LAYER_MBS") != NULL ? WELS_MAX (0, atoi (getenv ("WELS_HIP_MIN_LAYER_MBS")))
                    : (pParam->iSpatialLayerNum > 1 && st->layer_devices == 0) ? 1000 : 0;      // (640x360 = 920 macroblocks and below; a GPU per layer: the caller placed them)
  st->downsample = !(getenv ("WELS_HIP_DOWNSAMPLE") != NULL && atoi (getenv ("WELS_HIP_DOWNSAMPLE")) == 0);
  st->vaa = !(getenv ("WELS_HIP_VAA") != NULL && atoi (getenv ("WELS_HIP_VAA")) == 0);
  st->vaa_check = getenv ("WELS_HIP_CHECK_VAA") != NULL && atoi (getenv ("WELS_HIP_CHECK_VAA")) != 0;
  st->bgd = st->vaa && !(getenv ("WELS_HIP_BGD") != NULL && atoi (getenv ("WELS_HIP_BGD")) == 0);
  st->bgd_check = getenv ("WELS_HIP_CHECK_BGD") != NULL && atoi (getenv ("WELS_HIP_CHECK_BGD")) != 0;
  pFuncList->pHipState = st;
  pFuncList->pfHipFrameMd = HipFrameMd;
  pFuncList->pfHipCodeSlice = HipCodeSlice;
  pFuncList->pfHipLayerOnDevice = HipLayerOnDevice;
  pFuncList->pfHipRelease = HipRelease;
  pFuncList->pfHipDownsample = HipDownsample;
  pFuncList->pfHipVaaCalc = HipVaaCalc;
  pFuncList->pfHipFetchRecon = HipFetchRecon;
  pFuncList->pfHipBgd = HipBgd;
  Report (pLogCtx, "installed (device %d, %d spatial layer(s), %s)", st->device, pParam->iSpatialLayerNum, pParam->iUsageType == SCREEN_CONTENT_REAL_TIME ? "screen content" : "camera video");
}

}  // namespace WelsEnc
#endif  // HAVE_HIP
