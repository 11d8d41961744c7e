/* welship.h -- C ABI of libwelship.so, the MI355X-native macroblock engine behind the OpenH264
 * encoder API.
 *
 * Three layers, each one the drop-in counterpart of a reference interface (paths relative to the
 * cisco/openh264 tree):
 *
 *  (1) SESSION  -- mirrors ISVCEncoder (codec/api/wels/codec_api.h:272-343; C vtable :477-493):
 *      WelsHipCreateEncoder/Destroy  <->  WelsCreateSVCEncoder / WelsDestroySVCEncoder (:545,:552)
 *      WelsHipInitializeExt          <->  ISVCEncoder::InitializeExt (const SEncParamExt*)   (:286)
 *      WelsHipGetDefaultParams       <->  ISVCEncoder::GetDefaultParams                      (:293)
 *      WelsHipEncodeFrame            <->  ISVCEncoder::EncodeFrame (SSourcePicture*, SFrameBSInfo*) (:307)
 *      WelsHipForceIntraFrame        <->  ISVCEncoder::ForceIntraFrame                       (:323)
 *      WelsHipEncodeParameterSets    <->  ISVCEncoder::EncodeParameterSets                   (:316)
 *      WelsHipSetOption / GetOption  <->  ISVCEncoder::SetOption / GetOption                 (:329,:337)
 *      WelsHipUninitialize           <->  ISVCEncoder::Uninitialize                          (:298)
 *      Same call order, same ownership rules (input planes are caller-owned for the duration of the
 *      call; output buffers are encoder-owned and valid until the next call on the same object),
 *      return 0 = cmResultSuccess, 1 = cmInitParaError, 2 = cmUnknownReason, 3 = cmMallocMemeError,
 *      4 = cmUnsupportedData (codec_def.h:80-87).  integration/welship_isvc.cpp is the ISVCEncoder
 *      class over these functions; the reference's console front-end runs on it unmodified.
 *
 *  (2) FRAME-LEVEL PHASES on device-resident pictures, batched over N sessions (WelsHipGroup*) -- what
 *      the patched reference reaches through its dispatch table instead of its per-MB loops:
 *      UploadSource -> Begin (frame type, incl. the scene-change pass) -> RunDevice -> Finish, where
 *      RunDevice is, per picture batch,
 *        mode decision + reconstruction   <->  WelsISliceMdEnc / WelsMdInterMbLoop
 *                      (codec/encoder/core/src/svc_encode_slice.cpp:534-599, :1807-1899)
 *        in-loop deblocking               <->  pfDeblocking.pfDeblockingFilterSlice / DeblockingFilterFrameAvcbase
 *                      (codec/encoder/core/inc/wels_func_ptr_def.h:86-101, deblocking.cpp:656-691)
 *        border expansion                 <->  pfExpandLumaPicture / pfExpandChromaPicture
 *                      (wels_func_ptr_def.h, codec/common/src/expand_pic.cpp:271-350)
 *      and Finish is the host's pfWelsSpatialWriteMbSyn loop over the downloaded MB records.
 *
 *  (3) LEAF PRIMITIVES, batched over arrays of blocks -- same per-block semantics as the entries of
 *      SWelsFuncPtrList (codec/encoder/core/inc/wels_func_ptr_def.h:58-296) and SMcFunc
 *      (codec/common/inc/mc.h:40-53); see the WelsHipPrim* declarations below.  The same primitives one call at a time,
 *      with exactly the reference's typedefs (installable into the table's slots): welship_leaf.h.
 *
 * No PyTorch types appear here: plain pointers and sizes only.  Pointers named d_* are DEVICE
 * (HBM) addresses, everything else is host memory.  The library needs an MI355X (gfx950); every
 * entry point that touches the device returns WELSHIP_ERR_NO_DEVICE when none is usable -- there
 * is no CPU fallback.
 */
#ifndef WELSHIP_H_
#define WELSHIP_H_
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WELSHIP_OK 0
#define WELSHIP_ERR_INIT_PARA 1      /* cmInitParaError   */
#define WELSHIP_ERR_UNKNOWN 2        /* cmUnknownReason   */
#define WELSHIP_ERR_MEMORY 3         /* cmMallocMemeError: also what the reference returns when a frame overflows its bitstream buffer */
#define WELSHIP_ERR_UNSUPPORTED 4    /* cmUnsupportedData */
#define WELSHIP_ERR_NO_DEVICE 100
#define WELSHIP_ERR_VLC_OVERFLOW 101 /* internal: a macroblock must be re-encoded at QP+2 (handled inside EncodeFrame / GroupFinish) */

/* ---- (1) session API ---------------------------------------------------------------------- */

/* Subset of SEncParamExt (codec_app_def.h:540-598) this engine honours; field names kept. */
typedef struct WelsHipEncParam {
  int32_t iUsageType;               /* 0 = CAMERA_VIDEO_REAL_TIME (only value supported)            */
  int32_t iPicWidth, iPicHeight;
  int32_t iTargetBitrate;           /* bps; only feeds level selection while iRCMode == -1          */
  int32_t iRCMode;                  /* -1 = RC_OFF_MODE (constant QP) is the only supported mode    */
  float   fMaxFrameRate;
  int32_t iTemporalLayerNum;        /* 1                                                            */
  int32_t iSpatialLayerNum;         /* 1                                                            */
  int32_t iComplexityMode;          /* 0 LOW (SAD / fast I4x4), 1 MEDIUM, 2 HIGH (SATD / full I4x4)  */
  uint32_t uiIntraPeriod;           /* 0 = first frame only, N = IDR every N frames                  */
  int32_t eSpsPpsIdStrategy;        /* 0 CONSTANT_ID, 1 INCREASING_ID, 2 SPS_LISTING, 3 SPS_LISTING_AND_PPS_INCREASING */
  int32_t iEntropyCodingModeFlag;   /* 0 = CAVLC (CABAC is host work that is not implemented yet)    */
  int32_t iLoopFilterDisableIdc;    /* 0, 1, 2                                                       */
  int32_t iLoopFilterAlphaC0Offset, iLoopFilterBetaOffset;
  int32_t bEnableFrameCroppingFlag;
  int32_t iDLayerQp;                /* sSpatialLayers[0].iDLayerQp                                   */
  int32_t uiSliceMode;              /* 0 SM_SINGLE_SLICE, 1 SM_FIXEDSLCNUM_SLICE, 2 SM_RASTER_SLICE  */
  int32_t uiSliceNum;
  /* bEnableSceneChangeDetect: implemented (IDR on a LARGE_CHANGED_SCENE, encoder.cpp:377-391).  bEnableAdaptiveQuant is
   * accepted and ignored like the reference does (encoder_ext.cpp:300-301); bEnableFrameSkip only acts under rate
   * control.  bEnableBackgroundDetection, bEnableLongTermReference and bEnableDenoise must be 0 (cmUnsupportedData). */
  int32_t bEnableAdaptiveQuant, bEnableBackgroundDetection, bEnableSceneChangeDetect,
          bEnableLongTermReference, bEnableDenoise, bEnableFrameSkip;
  int32_t iDevice;                  /* HIP device ordinal                                            */
  int32_t iMultipleThreadIdc;       /* as in SEncParamExt (0 auto, 1 off, >1 threads); 0 is read as 1 -- see below.  This engine
                                       has no host slice threads; the field only reproduces what the reference does to the
                                       STREAM when it runs slice threads: deblocking across slice edges is switched off
                                       (idc 0 -> 2) because slices are filtered concurrently (encoder_ext.cpp:2051-2055)   */
  int32_t reserved[6];
  uint32_t uiSliceMbNum[35];        /* SM_RASTER_SLICE: macroblocks per slice (sSliceArgument.uiSliceMbNum, MAX_SLICES_NUM
                                       entries); uiSliceMbNum[0] == 0 = one slice per macroblock row                  */
} WelsHipEncParam;

/* SSourcePicture (codec_app_def.h:659-671), I420 only */
typedef struct WelsHipSourcePicture {
  int32_t iColorFormat;             /* 23 = videoFormatI420 */
  int32_t iStride[4];
  const uint8_t* pData[4];
  int32_t iPicWidth, iPicHeight;
  int64_t uiTimeStamp;
} WelsHipSourcePicture;

/* SLayerBSInfo / SFrameBSInfo (codec_app_def.h:621-657) */
#define WELSHIP_MAX_LAYER_NUM_OF_FRAME 128
#define WELSHIP_NON_VIDEO_CODING_LAYER 0
#define WELSHIP_VIDEO_CODING_LAYER 1
enum { WelsHipFrameTypeInvalid = 0, WelsHipFrameTypeIDR = 1, WelsHipFrameTypeI = 2, WelsHipFrameTypeP = 3, WelsHipFrameTypeSkip = 4 };
typedef struct WelsHipLayerBSInfo {
  uint8_t uiTemporalId, uiSpatialId, uiQualityId;
  int32_t eFrameType;
  uint8_t uiLayerType;
  int32_t iSubSeqId;
  int32_t iNalCount;
  int32_t* pNalLengthInByte;
  uint8_t* pBsBuf;
} WelsHipLayerBSInfo;
typedef struct WelsHipFrameBSInfo {
  int32_t iLayerNum;
  WelsHipLayerBSInfo sLayerInfo[4];
  int32_t eFrameType;
  int32_t iFrameSizeInBytes;
  int64_t uiTimeStamp;
} WelsHipFrameBSInfo;

typedef struct WelsHipEncoder WelsHipEncoder;

int  WelsHipCreateEncoder (WelsHipEncoder** ppEncoder);
void WelsHipDestroyEncoder (WelsHipEncoder* pEncoder);
int  WelsHipGetDefaultParams (WelsHipEncoder* pEncoder, WelsHipEncParam* pParam);
int  WelsHipInitializeExt (WelsHipEncoder* pEncoder, const WelsHipEncParam* pParam);
int  WelsHipUninitialize (WelsHipEncoder* pEncoder);
int  WelsHipEncodeFrame (WelsHipEncoder* pEncoder, const WelsHipSourcePicture* kpSrcPic, WelsHipFrameBSInfo* pBsInfo);
int  WelsHipForceIntraFrame (WelsHipEncoder* pEncoder, int bIDR);
/* ISVCEncoder::EncodeParameterSets (codec_api.h:316): SPS + PPS alone, one non-VCL layer with two NALs */
int  WelsHipEncodeParameterSets (WelsHipEncoder* pEncoder, WelsHipFrameBSInfo* pBsInfo);
/* Test/diagnostic hook (the reference's -drec / DumpDependencyRec, encoder_ext.cpp:3909-3915):
 * copies the last reconstructed (deblocked) frame, cropped to iPicWidth x iPicHeight, as I420. */
int  WelsHipGetReconFrame (WelsHipEncoder* pEncoder, uint8_t* pDstI420, size_t uiDstBytes);
/* ISVCEncoder::SetOption / GetOption (codec_api.h:329-337) for the options that mean something without rate control;
 * eOptionId takes the ENCODER_OPTION values of codec_app_def.h:106-146.  Implemented: ENCODER_OPTION_DATAFORMAT (0, int:
 * only videoFormatI420), ENCODER_OPTION_IDR_INTERVAL (1, int; <= -1 means 0), ENCODER_OPTION_FRAME_RATE (4, float; get
 * and set, no effect on the stream while RC is off), ENCODER_OPTION_COMPLEXITY (int, effective from the next picture),
 * ENCODER_OPTION_TRACE_LEVEL / TRACE_CALLBACK / TRACE_CALLBACK_CONTEXT (accepted, ignored).  Others: cmUnsupportedData
 * -- except that the reference's SetOption returns cmInitParaError for ids it does not know. */
#define WELSHIP_OPTION_DATAFORMAT 0
#define WELSHIP_OPTION_IDR_INTERVAL 1
#define WELSHIP_OPTION_FRAME_RATE 4
#define WELSHIP_OPTION_COMPLEXITY 15
#define WELSHIP_OPTION_TRACE_LEVEL 21
#define WELSHIP_OPTION_TRACE_CALLBACK 22
#define WELSHIP_OPTION_TRACE_CALLBACK_CONTEXT 23
int  WelsHipSetOption (WelsHipEncoder* pEncoder, int eOptionId, void* pOption);
int  WelsHipGetOption (WelsHipEncoder* pEncoder, int eOptionId, void* pOption);
/* Name of the device backend in use ("hip:gfx950 ..."). */
const char* WelsHipBackendName (WelsHipEncoder* pEncoder);
const char* WelsHipGetLastError (void);

/* ---- (1b) session group: N independent sessions with identical parameters advancing in lock
 * step, one batched launch set per frame step.  This is the multi-session form of the call stack
 * above (the reference runs N ISVCEncoder objects on N threads); it exists because one
 * picture exposes at most mb_w/2 independent macroblocks while an MI355X wants thousands. ------- */
typedef struct WelsHipEncoderGroup WelsHipEncoderGroup;
int  WelsHipGroupCreate (WelsHipEncoderGroup** ppGroup, const WelsHipEncParam* pParam, int iSessions,
                         int iSourceRingSlots, int iHostEntropyThreads);
void WelsHipGroupDestroy (WelsHipEncoderGroup* pGroup);
/* one EncodeFrame for every session: kpSrcPics[iSessions] -> pBsInfos[iSessions] */
int  WelsHipGroupEncodeFrames (WelsHipEncoderGroup* pGroup, const WelsHipSourcePicture* kpSrcPics, WelsHipFrameBSInfo* pBsInfos);
/* The same step as a software pipeline: the call SUBMITS kpSrcPics (staging copies and H2D on worker threads, then the kernels: all
 * queued) and meanwhile FINISHES the oldest pending step on a second thread (D2H of its packed records, CAVLC + NAL packing), so the
 * bitstreams come back iStepsAhead calls late: *pbFinished = 1 when pBsInfos[] holds that earlier step's output.  kpSrcPics = NULL only
 * finishes the oldest pending step (end of the streams: call until *pbFinished stays 0).  The streams are byte-identical to
 * WelsHipGroupEncodeFrames' (ISVCEncoder::EncodeFrame per session, codec/api/wels/codec_api.h:343), a CAVLC overflow found late
 * included: that picture and the ones submitted after it are coded again.  Scene-change detection is not available in this mode (it
 * needs a device statistic before every picture's type is decided).  WelsHipGroupSetPipelined (iStepsAhead = 1..3) before the first
 * picture: every step ahead costs each session one more reconstruction picture, record buffer and staging buffer. */
int  WelsHipGroupSetPipelined (WelsHipEncoderGroup* pGroup, int iStepsAhead);
int  WelsHipGroupEncodeFramesPipelined (WelsHipEncoderGroup* pGroup, const WelsHipSourcePicture* kpSrcPics, WelsHipFrameBSInfo* pBsInfos, int* pbFinished);
/* the same, split into its phases (sources may be made resident in HBM ahead of time).  iSlot selects one of the
 * iSourceRingSlots (at least 2) resident source pictures per session; a P picture must not use the slot of the picture
 * before it, whose source LOW complexity mode still reads (cmInitParaError otherwise). */
int  WelsHipGroupUploadSource (WelsHipEncoderGroup* pGroup, int iSession, int iSlot, const WelsHipSourcePicture* kpSrcPic);
int  WelsHipGroupBegin (WelsHipEncoderGroup* pGroup, int iSlot);
int  WelsHipGroupRunDevice (WelsHipEncoderGroup* pGroup, int bWait);
int  WelsHipGroupFinish (WelsHipEncoderGroup* pGroup, WelsHipFrameBSInfo* pBsInfos);
int  WelsHipGroupStepDeviceOnly (WelsHipEncoderGroup* pGroup, int iSlot);
int  WelsHipGroupGetReconFrame (WelsHipEncoderGroup* pGroup, int iSession, uint8_t* pDstI420, size_t uiDstBytes);
const char* WelsHipGroupBackendName (WelsHipEncoderGroup* pGroup);
/* hot-path timing with HIP events on the launch stream; pOutMs[4] = total, MD, deblock, expand */
int  WelsHipGroupBench (WelsHipEncoderGroup* pGroup, int iSteps, int iWarmup, double* pOutMs);
/* host share of the complete frame steps (EncodeFrames / Finish) so far, thread time per picture: pOut[4] = staging copy of
 * the source into page-locked memory (ms), entropy coding + NAL packing from the packed records (ms), pictures coded,
 * packed record bytes copied back per picture */
int  WelsHipGroupHostStats (WelsHipEncoderGroup* pGroup, double* pOut);
/* developer aid: WhMbRecord[] (openh264_amd/csrc/common/wh_types.h) of the last encoded frame */
int  WelsHipDebugGetMbRecords (WelsHipEncoder* pEncoder, void* pDst, size_t uiBytes);
/* developer aid: number of picture re-encodes caused by CAVLC level overflows since InitializeExt (the reference's
 * TRY_REENCODING loop, codec/encoder/core/src/svc_encode_slice.cpp:572-576,1863-1867), or -1 */
int  WelsHipDebugGetOverflowReencodes (WelsHipEncoder* pEncoder);
/* developer aid: the device's macroblock processing order of the MB range [iFirstMb, iLastMb) of a picture iMbWidth wide
 * (openh264_amd/csrc/common/mb_order.h; iBand = rows per band, 0 = the default single band); pOut: iLastMb - iFirstMb entries */
int  WelsHipDebugBuildMbOrder (int iMbWidth, int iFirstMb, int iLastMb, int iBand, uint16_t* pOut);
/* developer aid: the whole-picture deblocking order as items of one or two macroblocks (mb_order.h wh_build_db_pair_items: pOut[0] = number of
 * items, pOut[1 + i] = the item's first macroblock A as x | y << 12, bit 31 set when the item is the pair A, A + iMbWidth - 2 -- two macroblocks of one
 * 2:1 diagonal that one wavefront filters at once; only on diagonals of at least iMinLen macroblocks); pOut: iMbWidth * iMbHeight + 1 entries */
int  WelsHipDebugBuildDbPairItems (int iMbWidth, int iMbHeight, int iMinLen, uint32_t* pOut);
/* developer aid: per-phase cycle counters accumulated inside the MB kernels; pOut64 receives 16 sums + 16 counts of the
 * mode-decision kernel followed by 16 sums + 16 counts of the deblocking kernel */
int  WelsHipGroupProfile (WelsHipEncoderGroup* pGroup, int bEnable, unsigned long long* pOut64);

/* ---- (2b) EXPLICIT FRAME API: what the SWelsFuncPtrList hooks of the patched reference call ---------------------
 * (integration/welship_hooks.cpp + integration/openh264_hip.patch).  The reference keeps its frame layer,
 * reference-list management (temporal layers, LTR), pre-processing, rate control and entropy coder; this library keeps a
 * device twin of each of the reference's reconstructed pictures (SPicture, codec/encoder/core/inc/picture.h:64-121) and does,
 * per picture, what WelsISliceMdEnc / WelsMdInterMbLoop (svc_encode_slice.cpp:534-599,1807-1899), PerformDeblockingFilter
 * (deblocking.cpp:744-762) and ExpandReferencingPicture (ref_list_mgr_svc.cpp:375) do on the CPU.  It hands back one
 * macroblock record per MB -- WhMbRecord, openh264_amd/csrc/common/wh_types.h: SMB + SMbCache side info and the SDCTCoeff
 * levels (mb_cache.h:62-70) in the reference's own order -- for the host's pfWelsSpatialWriteMbSyn loop. */
typedef struct WelsHipFrameCtx WelsHipFrameCtx;
typedef struct WelsHipFrameCfg {
  int32_t iDevice;
  int32_t iPicWidth, iPicHeight;    /* luma samples; coded size = rounded up to whole macroblocks                              */
  int32_t iNumPictures;             /* device pictures: the reference pool + the picture being coded (AllocPicture count)      */
  int32_t reserved[4];
} WelsHipFrameCfg;
/* Screen content (iUsageType == SCREEN_CONTENT_REAL_TIME), P pictures: what the reference's pre-processing and PreprocessSliceCoding
 * (encoder_ext.cpp:2700-2765) hand to WelsMdInterJudgeSCDPskip / WelsMdInterFinePartitionVaaOnScreen (svc_mode_decision.cpp:326-667)
 * and to the cross / feature searches (svc_motion_estimate.cpp:380-1097).  All pointers are host memory of the caller. */
typedef struct WelsHipScreenInfo {
  const uint8_t* pBlockStaticIdc;       /* pVaaExt->pVaaBestBlockStaticIdc: EStaticBlockIdc per 8x8 luma block, [2 * mb_h][2 * mb_w]          */
  const uint8_t* pRefOriChroma[2];      /* pCurDqLayer->pRefOri[0]->pData[1], [2]: chroma of the reference picture's SOURCE, or NULL          */
  int32_t iRefOriStride;
  int32_t bScrollDetectFlag, iScrollMvX, iScrollMvY;   /* pVaaExt->sScrollDetectInfo                                                         */
  uint32_t uiSadCostThreshold16x16, uiSadCostThreshold8x8;   /* pRefPic->pScreenBlockFeatureStorage->uiSadCostThreshold[BLOCK_16x16 / _8x8]  */
  int32_t bFeatureSearch8x8;            /* pfSearchMethod[BLOCK_8x8] == WelsDiamondCrossFeatureSearch for this picture                       */
  int32_t bStaticSkipDecision;          /* pfSCDPSkipDecision == WelsMdInterJudgeSCDPskip (encoder.cpp:205-208: off with HIGH complexity)    */
  /* the reference picture's SScreenBlockFeatureStorage as PerformFMEPreprocess built it (svc_motion_estimate.cpp:839-873); read only   */
  /* with bFeatureSearch8x8                                                                                                              */
  const uint32_t* pTimesOfFeatureValue; /* [iListSize]                                                                                       */
  uint16_t* const* pLocationOfFeature;  /* [iListSize] pointers into pLocationPointer                                                        */
  const uint16_t* pLocationPointer;     /* {x << 2, y << 2} of every 8x8 block position, iLocationEntries entries                            */
  int32_t iListSize, iLocationEntries;
  uint32_t* pSliceFMECostDown;          /* out, [iNumSlices]: what the picture adds to each pSlice->uiSliceFMECostDown                       */
} WelsHipScreenInfo;
typedef struct WelsHipFrameJob {
  uint32_t cbSize;                  /* sizeof (WelsHipFrameJob) of the header the caller was compiled with.  New fields are only ever APPENDED, */
                                    /* and zero means "not used": a caller built against an older header (cbSize from WELSHIP_FRAMEJOB_MIN_SIZE, */
                                    /* the first layout that carried this field, up to the library's own sizeof) is served with the fields it   */
                                    /* does not know taken as zero; anything shorter, or longer than the library's struct (a caller newer than  */
                                    /* the library, whose extra fields it could not honour), is refused with WELSHIP_ERR_INIT_PARA             */
  int32_t iCurPic, iRefPic;         /* device picture indices (0 .. iNumPictures-1); iRefPic < 0: I picture                    */
  int32_t eSliceType;               /* 0 = P_SLICE, 2 = I_SLICE (slice_type values of the standard)                            */
  int32_t iQp;                      /* pEncCtx->iGlobalQp: WelsRcMbInitDisable / WelsRcMbInitGom with bEnableGomQp == false    */
  int32_t iChromaQpIndexOffset;     /* pPps->uiChromaQpIndexOffset                                                             */
  int32_t iComplexityMode;          /* pSvcParam->iComplexityMode as PreprocessSliceCoding reads it (encoder_ext.cpp:2658)     */
  int32_t iMvRange;                 /* pEncCtx->iMvRange                                                                       */
  int32_t iMvcShift;                /* pSlice->sScaleShift (svc_encode_slice.cpp:1652-1655)                                    */
  int32_t iNumSlices;               /* slices are contiguous MB ranges: pSliceFirstMb[iNumSlices + 1]                           */
  const int32_t* pSliceFirstMb;
  int32_t iDeblockIdc, iAlphaOffset, iBetaOffset;   /* slice-header values of this picture (offsets as coded, i.e. x2)        */
  int32_t bDeblock;                 /* run the in-loop filter (encoder_ext.cpp:3870-3880: not for the highest temporal layer)  */
  int32_t bExpand;                  /* the picture enters the reference list: replicate its borders                            */
  const uint8_t* pSrc[3];           /* pEncPic planes (host); the MB-aligned area must be readable (the reference pads it)     */
  int32_t iSrcStride[3];
  const int32_t* pVaaSad8x8;        /* pVaa->sVaaCalcInfo.pSad8x8 ([mb][4], host) or NULL (then LOW complexity is refused)     */
  const int8_t* pBgdFlags;          /* pVaa->pVaaBackgroundMbFlag (host) or NULL: background detection off                     */
  const uint8_t* pMbQp;             /* per-MB luma QP (host, [mb]) or NULL: iQp for every MB (GOM-level rate control uses it)  */
  int32_t iMbBegin, iMbEnd;         /* code only this MB range now (GOM-synchronous rate control); 0,0 = the whole picture.     */
                                    /* Deblocking / expansion run with the call whose range ends the picture.                  */
  const int16_t* pIlHint;           /* highest spatial layer of a multi-layer session: per MB {sMvBase x, y, bit 0 = the layer below  */
                                    /* is intra there, 0} as SetMvBaseEnhancelayer / GetRefMb derive them (svc_mode_decision.cpp:     */
                                    /* 108-150), or NULL                                                                               */
  int32_t* pSadCost;                /* pSadCost[0] of every MB (pEncCtx->pSadCostMb, encoder_ext.cpp:900,1675 -- ONE array for all the    */
                                    /* spatial layers of a session): copied to the device before the picture and back after it, or NULL:  */
                                    /* the context keeps its own (single-layer sessions)                                                 */
  const WelsHipScreenInfo* pScreen; /* screen-content P pictures, else NULL (I pictures of a screen-content session: iComplexityMode >= 1, */
                                    /* PreprocessSliceCoding selects the SATD / full-search intra functions for them)                       */
  /* TRY_REENCODING (svc_encode_slice.cpp:564-576,1845-1867): the entropy writer could not code a macroblock (CAVLC level overflow,  */
  /* or the picture's bitstream buffer nearly full) and the reference codes it again with its QP raised by 2.  The caller repeats   */
  /* the call with bRetry = 1 and the list of ALL macroblocks re-encoded so far in this picture: the device inputs uploaded by the   */
  /* first call are reused (the host may have modified its copies meanwhile), every other macroblock reproduces itself.              */
  int32_t bRetry;
  int32_t bCountBits;               /* WhMbRecord::cavlc_bits of every macroblock: the bits its CAVLC syntax will take, counted on the   */
                                    /* device (set_mb_syn_cavlc.cpp:84-232, svc_set_mb_syn_cavlc.cpp:58-440), for GOM-level rate control  */
  int32_t iNumReencode;
  int32_t iNumRefIdxL0Active;       /* P pictures with bCountBits: the slice header's num_ref_idx_l0_active (pEncCtx->iNumRef0)          */
  const struct WelsHipGomRc* pGomRc; /* P pictures of one slice under rate control (GOM-level QP, ratectl.cpp:1239-1278): the whole picture */
                                    /* in ONE call -- the device counts the bits of every group of macroblocks and runs RcCalculateGomQp /  */
                                    /* RcGomTargetBits between the groups itself (iQp = the first group's QP, pMbQp / iMbBegin unused);     */
                                    /* the records carry each macroblock's QP, which the caller's own rate control must arrive at too       */
  const struct WelsHipMbReencode* pReencode;
  /* Size-limited slices (SM_SIZELIMITED_SLICE, uiSliceSizeConstraint): where a slice ends is only known once the entropy writer has   */
  /* produced its bytes (DynSlcJudgeSliceBoundaryStepBack, svc_encode_slice.cpp:1741-1790), and the macroblock a new slice begins   */
  /* with is decided again without its neighbours (WelsMdInterMbLoopOverDynamicSlice :1901-2010, WelsISliceMdEncDynamic :601-680).    */
  /* iDynSlice = 1 + index of the slice the macroblocks [iMbBegin, iMbEnd) belong to, which begins at iDynSliceFirstMb (<= iMbBegin):  */
  /* the device codes them AHEAD of the writer, as if the slice went on to iMbEnd; when the writer ends the slice at macroblock b the    */
  /* caller repeats the call for [b, ...) as the next slice and everything from b on is coded again.  The slice table of such a       */
  /* picture are its PARTITIONS (one per slice thread, each sliced on its own; one thread: the whole picture) and a range stays        */
  /* inside one; calls for different partitions may come from different threads.  The picture-wide passes wait for a closing call     */
  /* with iMbBegin = iMbEnd = the number of macroblocks (nothing is coded by it).  0 = slices as in pSliceFirstMb.                     */
  int32_t iDynSlice, iDynSliceFirstMb;
  /* A macroblock of an MB range overflowed (TRY_REENCODING above): the call is repeated for the rest of the range, from that macroblock   */
  /* on, with pReencode / iNumReencode (bRetry stays 0) -- and every later call of the picture, the closing one included, carries the list. */
  /* bRangeAgain = 1 marks such a repeated call: even when it begins at macroblock 0 the picture's inputs are not uploaded again.           */
  int32_t bRangeAgain;
  /* Size-limited slices: the range's first macroblock is decided for the second time in this picture -- the slice begins with the        */
  /* macroblock the writer took back, or bRangeAgain -- and finds what its first pass left in the layer's pSadCost array                  */
  /* (WelsMdInterSaveSadAndRefMbType runs before the writer), every other macroblock what the previous picture left.                      */
  int32_t bDynRedoFirst;
  /* Whole-picture calls: hand the records over PACKED, as they crossed PCIe (a skipped macroblock 16 bytes, any other its 144 bytes of side     */
  /* information plus the 32-byte level blocks that hold a level the entropy coder can read: openh264_amd/csrc/common/compact.h) -- *ppRecords   */
  /* then points at a WelsHipPackedRecords instead of a WhMbRecord array, and the caller expands a macroblock when it gets to it                  */
  /* (wh_compact_expand).  About a sixth of the 960-byte records on camera content; MB ranges always come back as full records.                   */
  /* Pictures of more than WELSHIP_PACKED_MAX_MB macroblocks (and a library run with WELSHIP_COMPACT=0) cannot be packed: the library says what  */
  /* *ppRecords holds through *pbRecordsPacked (1 = a WelsHipPackedRecords, 0 = the WhMbRecord array), which a caller that sets bPackedRecords     */
  /* MUST supply -- a request without it is refused (WELSHIP_ERR_INIT_PARA), never answered in a format the caller cannot tell.                    */
  int32_t bPackedRecords;
  int32_t* pbRecordsPacked;
} WelsHipFrameJob;
#define WELSHIP_FRAMEJOB_MIN_SIZE 232u      /* sizeof (WelsHipFrameJob) of the first layout with cbSize (LP64): up to and including pbRecordsPacked */
#define WELSHIP_PACKED_MAX_MB 9216
typedef struct WelsHipPackedRecords {
  const uint8_t* pData;             /* the packed stream of the picture                                                                       */
  const uint32_t* pOffset;          /* [number of macroblocks + 1] byte offsets into pData; macroblock mb is pOffset[mb + 1] - pOffset[mb] long */
} WelsHipPackedRecords;
typedef struct WelsHipGomRc {
  int32_t iNumberMbGom;             /* pWelsSvcRc->iNumberMbGom: whole macroblock rows (else WELSHIP_ERR_UNSUPPORTED: code the groups one by one) */
  int32_t iEndMbSlice, iTargetBitsSlice;      /* pSlice->sSlicingOverRc                                                                  */
  int32_t iMinFrameQp, iMaxFrameQp; /* pWelsSvcRc                                                                                        */
  int32_t iGomSize;
  const int32_t* pGomSad;           /* pCurrentFrameGomSad[iGomSize] of the layer RcGomTargetBits reads (RcJudgeBaseUsability)           */
} WelsHipGomRc;
typedef struct WelsHipMbReencode {
  int32_t iMbXY;
  uint8_t uiLumaQp;                 /* the QP of this pass (UpdateQpForOverflow)                                                       */
  uint8_t uiStaleCbp;               /* uiCbp the previous pass left (only WelsMdIntraInit clears it, and that is not repeated)         */
  uint8_t bCell12Valid, pad;        /* the previous pass ended as P8x16: update_P8x16_motion_info wrote its second vector into the     */
  int16_t iCell12Mv[2];             /* MV cache's left-neighbour cell (mv_pred.cpp:235-276); the vector                                */
} WelsHipMbReencode;
/* Pre-analysis of a source picture against an earlier one: what CWelsPreProcess::VaaCalculation (wels_preprocess.cpp:
 * 677-711) gets from CVAACalculation::Process (codec/processing/src/vaacalc/vaacalculation.cpp:118-157) -- VAACalcSad_c /
 * VAACalcSadBgd_c / VAACalcSadSsd_c / VAACalcSadVar_c / VAACalcSadSsdBgd_c (vaacalcfuncs.cpp:37-600), selected by the three flags --
 * computed on the device.  The CURRENT picture is uploaded by this call and stays resident: the WelsHipFrameEncode of the same
 * context that follows with the same pSrc[0] does not upload it again; the EARLIER picture is found on the device when it was the
 * source of an earlier call of this context (keyed by its luma pointer: the reference rotates a fixed set of picture buffers), and is
 * uploaded otherwise.  Results go to the caller's arrays (those the selected variant writes; macroblocks outside
 * (iPicWidth >> 4) x (iPicHeight >> 4) are left alone, as the C functions leave them). */
/* Widths that are no multiple of 16: the C functions walk such a picture skewed (every macroblock row begins (width & 15) samples further left,
 * vaacalcfuncs.cpp:46,145-146) -- pCur[0] / pRef[0] must then be readable for 16 * (iPicHeight >> 4) whole lines of iCurStride[0] bytes, and the two
 * strides must be equal (the C functions take one).  The picture is not left on the device in that case. */
typedef struct WelsHipVaaJob {
  const uint8_t* pCur[3]; int32_t iCurStride[3];      /* host planes of the current source picture, MB-aligned area readable    */
  const uint8_t* pRef[3]; int32_t iRefStride[3];      /* ... of the picture it is compared with                                  */
  int32_t iPicWidth, iPicHeight;                      /* sRect of the SPixMap the reference passes                               */
  int32_t bCalcVar, bCalcBgd, bCalcSsd;               /* SVAACalcParam::iCalcVar / iCalcBgd / iCalcSsd                           */
  int32_t* pSad8x8;                                   /* SVAACalcResult: [mb][4]                                                 */
  int32_t* pSsd16x16; int32_t* pSum16x16; int32_t* pSumOfSquare16x16;     /* [mb]                                                */
  int32_t* pSumOfDiff8x8; uint8_t* pMad8x8;           /* [mb][4]                                                                 */
  int32_t* pFrameSad;
} WelsHipVaaJob;
int  WelsHipFrameVaa (WelsHipFrameCtx* pCtx, const WelsHipVaaJob* pJob);
/* Background detection (CBackgroundDetection::Process, codec/processing/src/backgrounddetection/BackgroundDetection.cpp:52-85: what
 * CWelsPreProcess::BackgroundDetection runs right after VaaCalculation, wels_preprocess.cpp:286-301,713-761) of the picture pair the LAST
 * WelsHipFrameVaa call of this context analysed with bCalcBgd: the statistics and the two pictures are still on the device.
 * pBackgroundMbFlag [((iPicWidth + 15) >> 4) per row] receives the flag of every macroblock of the (iPicWidth >> 4) x (iPicHeight >> 4)
 * the reference covers; the others are left alone.  WELSHIP_ERR_UNSUPPORTED (the caller runs its C function) when the last call was for
 * another pair, without bCalcBgd, or for a width that is no multiple of 16. */
typedef struct WelsHipBgdJob {
  const uint8_t* pCur[3]; const uint8_t* pRef[3];     /* the planes handed to that WelsHipFrameVaa call (identify the pair; not read)    */
  int32_t iPicWidth, iPicHeight;
  int8_t* pBackgroundMbFlag;
} WelsHipBgdJob;
int  WelsHipFrameBgd (WelsHipFrameCtx* pCtx, const WelsHipBgdJob* pJob);
int  WelsHipFrameCtxCreate (WelsHipFrameCtx** ppCtx, const WelsHipFrameCfg* pCfg);
void WelsHipFrameCtxDestroy (WelsHipFrameCtx* pCtx);
/* Runs the picture (or MB range) on the device and waits; *ppRecords = WhMbRecord[mb_w * mb_h] in host memory, valid until
 * the next call (entries outside the coded range keep what earlier calls for the same picture produced). */
int  WelsHipFrameEncode (WelsHipFrameCtx* pCtx, const WelsHipFrameJob* pJob, const void** ppRecords);
/* The (deblocked) reconstruction of device picture iPic, coded size, into the caller's planes (the host's SPicture). */
int  WelsHipFrameGetPicture (WelsHipFrameCtx* pCtx, int iPic, uint8_t* const pDst[3], const int32_t iDstStride[3]);
/* The per-MB states of device picture iPic (WhMbState[mb_w * mb_h], wh_types.h: types, motion vectors, reference indices):
 * what a higher spatial layer's mode decision reads from this layer's SMB array. */
int  WelsHipFrameGetMbStates (WelsHipFrameCtx* pCtx, int iPic, void* pDst, size_t uiBytes);

/* ---- (3) leaf primitives, batched.  One call = n independent invocations of the reference entry
 * named in the comment.  p*Plane are HOST buffers of `bytes` bytes; block i starts at
 * plane + pOff[i] with the given stride (exactly the (pointer, stride) pairs the reference passes).
 * iBlock: BLOCK_16x16..BLOCK_4x8 = 0..6 (codec/encoder/core/inc/wels_const.h:139-148). -------- */
/* pfSampleSad[iBlock] / pfSampleSatd[iBlock] / pfSample4Sad[iBlock] (wels_func_ptr_def.h:162-177) */
int WelsHipPrimSampleSad (int iBlock, int n, const uint8_t* pPlane1, size_t bytes1, int32_t iStride1, const int32_t* pOff1,
                          const uint8_t* pPlane2, size_t bytes2, int32_t iStride2, const int32_t* pOff2, int32_t* pSad);
int WelsHipPrimSampleSatd (int iBlock, int n, const uint8_t* pPlane1, size_t bytes1, int32_t iStride1, const int32_t* pOff1,
                           const uint8_t* pPlane2, size_t bytes2, int32_t iStride2, const int32_t* pOff2, int32_t* pSatd);
int WelsHipPrimSample4Sad (int iBlock, int n, const uint8_t* pPlane1, size_t bytes1, int32_t iStride1, const int32_t* pOff1,
                           const uint8_t* pPlane2, size_t bytes2, int32_t iStride2, const int32_t* pOff2, int32_t* pSad4 /*[n][4] up,down,left,right*/);
/* pfDctT4 (PDctFunc): pDct[n][16] = T(pix1 - pix2) */
int WelsHipPrimDctT4 (int n, const uint8_t* pPlane1, size_t bytes1, int32_t iStride1, const int32_t* pOff1,
                      const uint8_t* pPlane2, size_t bytes2, int32_t iStride2, const int32_t* pOff2, int16_t* pDct);
/* pfQuantizationFour4x4Max (per 4x4) + pfScan4x4 + pfScan4x4Ac + pfCalculateSingleCtr4x4 + pfGetNoneZeroCount */
int WelsHipPrimQuant4x4 (int n, int16_t* pDctInOut /*[n][16]*/, const uint8_t* pQp, int bIntra, int16_t* pMax, int16_t* pScanDcAc,
                         int16_t* pScanAc, int32_t* pSingleCtr, int32_t* pNzc);
/* pfDequantization4x4 + pfIDctT4: pRec[n][16] = clip(pPred + idct(levels * dequant)) */
int WelsHipPrimDequantIDctRec (int n, const int16_t* pLevelsRaster, const uint8_t* pQp, const uint8_t* pPred, uint8_t* pRec, int16_t* pDequant);
/* pfGetLumaI4x4Pred: pMode = Intra4x4PredMode 0..8, pAvail bit0 left / bit1 top selects the DC flavour */
int WelsHipPrimIntraPred4x4 (int n, const uint8_t* pPlane, size_t bytes, int32_t iStride, const int32_t* pOff, const uint8_t* pMode,
                             const uint8_t* pAvail, uint8_t* pPred /*[n][16]*/);
/* pfGetLumaI16x16Pred[pMode16] + pfGetChromaPred[pModeChroma] (reference mode numbering incl. DC_L/DC_T/DC_128);
 * Cr samples are read 16 columns to the right of the Cb block in pPlaneC */
int WelsHipPrimIntraPredMb (int n, const uint8_t* pPlaneY, size_t bytesY, int32_t iStrideY, const int32_t* pOffY,
                            const uint8_t* pPlaneC, size_t bytesC, int32_t iStrideC, const int32_t* pOffC,
                            const uint8_t* pMode16, const uint8_t* pModeChroma, uint8_t* pPred16 /*[n][256]*/, uint8_t* pPredChroma /*[n][128]*/);
/* sMcFuncs.pMcLumaFunc / pMcChromaFunc: pOff addresses the integer sample position, pMv[n][2] supplies the fraction */
int WelsHipPrimMc (int n, const uint8_t* pPlane, size_t bytes, int32_t iStride, const int32_t* pOff, const int16_t* pMv, int iWidth,
                   int iHeight, int bChroma, uint8_t* pDst /*[n][h][w]*/);
/* pfLumaDeblocking{LT4,EQ4}{Ver,Hor} / pfChromaDeblocking*: edges at plane + pOff[e] (first q0 sample), bS per 4 (luma) or
 * 2 (chroma) lines in pBs4[e][4], alpha/beta/tc0 looked up at pIndexA[e] (slice offsets 0) */
int WelsHipPrimDeblockEdges (int nEdges, uint8_t* pPlaneInOut, size_t bytes, int32_t iStride, const int32_t* pOff, int bVerticalEdge,
                             int bChroma, const uint8_t* pBs4, const uint8_t* pIndexA);
/* VAACalcSad_c of codec/processing (per macroblock): four 8x8 SADs against the previous source picture */
int WelsHipPrimVaaSad8x8 (int nMb, const uint8_t* pCur, const uint8_t* pRef, size_t bytes, int32_t iStride, const int32_t* pOff, int32_t* pSad8x8);

/* ---- (3b) spatial down-sampling of source planes (simulcast layers; codec/processing/src/downsample): the entries of
 * SDownsampleFuncs as the C build fills them (downsample.cpp:73-93): pfHalfAverageWidthx32/x16 = DyadicBilinearDownsampler_c,
 * pfQuarterDownsampler, pfOneThirdDownsampler, pfGeneralRatioLuma = GeneralBilinearFastDownsampler_c, pfGeneralRatioChroma =
 * GeneralBilinearAccurateDownsampler_c (downsamplefuncs.cpp:47-245).  Host planes; the source plane must have one readable
 * row below its last one (the reference's planes are padded), destination samples outside iDstWidth x iDstHeight are untouched. */
#define WELSHIP_DS_HALF 0
#define WELSHIP_DS_QUARTER 1
#define WELSHIP_DS_ONE_THIRD 2
#define WELSHIP_DS_GENERAL_FAST 3
#define WELSHIP_DS_GENERAL_ACCURATE 4
int WelsHipPrimDownsample (int mode, uint8_t* pDst, int32_t iDstStride, int32_t iDstWidth, int32_t iDstHeight,
                           const uint8_t* pSrc, int32_t iSrcStride, int32_t iSrcWidth, int32_t iSrcHeight);
/* One picture, three planes, the method CDownsampling::Process (codec/processing/src/downsample/downsample.cpp:144-277) picks for the
 * size pair -- the dyadic cascade included -- in one call: what the dispatch-table binding runs in place of
 * m_pInterfaceVp->Process (METHOD_DOWNSAMPLE, ..) in CWelsPreProcess::DownsamplePadding (wels_preprocess.cpp:625-675; the padding
 * that follows stays the caller's).  Host planes in, host planes out (I420; chroma planes are (w >> 1) x (h >> 1)); samples outside
 * iDstWidth x iDstHeight are untouched.  Thread-safe; per device one queue with page-locked staging that is kept between calls. */
int WelsHipDownsamplePicture (int iDevice, uint8_t* const pDst[3], const int32_t iDstStride[3], int32_t iDstWidth, int32_t iDstHeight,
                              const uint8_t* const pSrc[3], const int32_t iSrcStride[3], int32_t iSrcWidth, int32_t iSrcHeight);
/* one launch over nPlanes HBM-resident planes, timed with HIP events: pOut[0] = ms per launch, pOut[1] = algorithmic bytes per launch */
int WelsHipDownsampleBench (int iDevice, int mode, int nPlanes, int iSrcWidth, int iSrcHeight, int iDstWidth, int iDstHeight, int iIters, double* pOut);

#ifdef __cplusplus
}
#endif
#endif  /* WELSHIP_H_ */
