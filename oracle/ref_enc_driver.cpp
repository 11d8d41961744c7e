// oracle/ref_enc_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A small command-line driver over the *reference's own* public encoder API
// (codec/api/wels/codec_api.h:272 ISVCEncoder, :545 WelsCreateSVCEncoder).  It links against
// oracle/_ref/libref_openh264.so (the reference compiled as-is by oracle/Makefile) and plays
// the role h264enc + welsenc.cfg/layer2.cfg play in the reference tree
// (codec/console/enc/src/welsenc.cpp), without needing those cfg files on the GPU box.
//
// Every option below maps 1:1 onto an SEncParamExt field (codec_app_def.h:540-598); defaults
// are GetDefaultParams() (param_svc.h:132 FillDefault) unless stated.
//
//   ref_enc -i in.yuv -w W -h H -o out.264 [-frames N] [-fps F]
//           [-rc M] [-qp Q] [-bitrate BPS] [-iper N] [-numtl N] [-complexity C]
//           [-slcmd M] [-slcnum N] [-slcmbnum N] [-slcsize BYTES] [-nalsize BYTES] [-threads N] [-loadbalancing 0/1]
//           [-deblock IDC] [-aq 0/1] [-bgd 0/1] [-scene 0/1] [-ltr 0/1] [-denoise 0/1]
//           [-frameskip 0/1] [-cabac 0/1] [-spsid S] [-usage U] [-base | -ext [-lossless 0/1]] [-quiet]
//
// Prints "frames=<n> bytes=<n> enc_seconds=<s> fps=<f>" (timed strictly around EncodeFrame,
// like welsenc.cpp:957-960).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <thread>
#include "codec_api.h"

static bool arg_eq (const char* a, const char* b) { return std::strcmp (a, b) == 0; }

// one encoder instance; `instance` >= 0: one of several running in this process at once (-parallel N), output file out.<instance>
static int run (int argc, char** argv, int instance) {
  std::string in, out;
  int w = 0, h = 0, frames = -1, quiet = 0, use_base = 0, use_ext = 0, lossless = 0, profile = 66;
  float fps = 30.0f;
  int rc = -1, qp = 24, bitrate = 5000000, iper = 0, numtl = 1, complexity = 0;
  int slcmd = 0, slcnum = 1, slcmbnum = 0, threads = 1, loadbal = 0, deblock = 0, slcsize = 0, nalsize = 0;
  int aq = 0, bgd = 0, scene = 0, ltr = 0, denoise = 0, frameskip = 0, cabac = 0, spsid = 1, usage = 0;
  int alpha = 0, beta = 0, crop = 1, forceidr = -1;
  int setidr_at = -1, setidr_val = 0, setcplx_at = -1, setcplx_val = 0, paramsets_at = -1, setfps_at = -1;
  float setfps_val = 30.f;
  int trace_level = -1;
  // re-configuration in mid-stream (welsEncoderExt.cpp:700-1100 SetOption; encoder_ext.cpp:4173 WelsEncoderParamAdjust): each before frame N
  int setbr_at = -1, setbr_val = 0;                 // -setbr N BPS: ENCODER_OPTION_BITRATE (SPATIAL_LAYER_ALL)
  int setres_at = -1, setres_w = 0, setres_h = 0;   // -setres N W H: ENCODER_OPTION_SVC_ENCODE_PARAM_EXT with another picture size; the input file holds W x H frames from frame N on
  int setltr_at = -1, setltr_val = 0;               // -setltr N 0/1: ENCODER_OPTION_LTR
  int ltrrecover_at = -1, ltrmarkfb_at = -1;        // -ltrrecover N / -ltrmarkfb N: ENCODER_LTR_RECOVERY_REQUEST / ENCODER_LTR_MARKING_FEEDBACK as a receiver would send them
  std::string infofile;                // -dumpinfo FILE: the SFrameBSInfo metadata of every frame, one line per layer
  int low_w = 0, low_h = 0;            // -simulcast W H: an extra, lower spatial layer, simulcast AVC (the input is the highest one);
  int lows[3][2] = {{0, 0}, {0, 0}, {0, 0}}, nlow = 0;     // may be given up to three times, lowest resolution first
  for (int i = 1; i < argc; ++i) {
    const char* a = argv[i];
    auto next = [&] () -> const char* { if (i + 1 >= argc) { std::fprintf (stderr, "missing value for %s\n", a); std::exit (2); } return argv[++i]; };
    if (arg_eq (a, "-i")) in = next();
    else if (arg_eq (a, "-o")) out = next();
    else if (arg_eq (a, "-w")) w = std::atoi (next());
    else if (arg_eq (a, "-h")) h = std::atoi (next());
    else if (arg_eq (a, "-frames")) frames = std::atoi (next());
    else if (arg_eq (a, "-fps")) fps = (float)std::atof (next());
    else if (arg_eq (a, "-rc")) rc = std::atoi (next());
    else if (arg_eq (a, "-qp")) qp = std::atoi (next());
    else if (arg_eq (a, "-bitrate")) bitrate = std::atoi (next());
    else if (arg_eq (a, "-iper")) iper = std::atoi (next());
    else if (arg_eq (a, "-numtl")) numtl = std::atoi (next());
    else if (arg_eq (a, "-complexity")) complexity = std::atoi (next());
    else if (arg_eq (a, "-slcmd")) slcmd = std::atoi (next());
    else if (arg_eq (a, "-slcnum")) slcnum = std::atoi (next());
    else if (arg_eq (a, "-slcmbnum")) slcmbnum = std::atoi (next());
    else if (arg_eq (a, "-slcsize")) slcsize = std::atoi (next());       // uiSliceSizeConstraint of -slcmd 3 (size-limited slices)
    else if (arg_eq (a, "-nalsize")) nalsize = std::atoi (next());       // uiMaxNalSize
    else if (arg_eq (a, "-threads")) threads = std::atoi (next());
    else if (arg_eq (a, "-loadbalancing")) loadbal = std::atoi (next());
    else if (arg_eq (a, "-deblock")) deblock = std::atoi (next());
    else if (arg_eq (a, "-aq")) aq = std::atoi (next());
    else if (arg_eq (a, "-bgd")) bgd = std::atoi (next());
    else if (arg_eq (a, "-scene")) scene = std::atoi (next());
    else if (arg_eq (a, "-ltr")) ltr = std::atoi (next());
    else if (arg_eq (a, "-denoise")) denoise = std::atoi (next());
    else if (arg_eq (a, "-frameskip")) frameskip = std::atoi (next());
    else if (arg_eq (a, "-cabac")) cabac = std::atoi (next());
    else if (arg_eq (a, "-spsid")) spsid = std::atoi (next());
    else if (arg_eq (a, "-usage")) usage = std::atoi (next());
    else if (arg_eq (a, "-base")) use_base = 1;
    else if (arg_eq (a, "-ext")) use_ext = 1;
    else if (arg_eq (a, "-profile")) profile = std::atoi (next());          // uiProfileIdc: 66 baseline (layer2.cfg), 77 main, 100 high (CABAC needs one of the latter)
    else if (arg_eq (a, "-lossless")) lossless = std::atoi (next());
    else if (arg_eq (a, "-alpha")) alpha = std::atoi (next());
    else if (arg_eq (a, "-beta")) beta = std::atoi (next());
    else if (arg_eq (a, "-crop")) crop = std::atoi (next());
    else if (arg_eq (a, "-forceidr")) forceidr = std::atoi (next());     // ForceIntraFrame(true) before frame N
    else if (arg_eq (a, "-setidr")) { setidr_at = std::atoi (next()); setidr_val = std::atoi (next()); }      // SetOption (ENCODER_OPTION_IDR_INTERVAL) before frame N
    else if (arg_eq (a, "-setcplx")) { setcplx_at = std::atoi (next()); setcplx_val = std::atoi (next()); }   // SetOption (ENCODER_OPTION_COMPLEXITY) before frame N
    else if (arg_eq (a, "-setfps")) { setfps_at = std::atoi (next()); setfps_val = (float)std::atof (next()); }  // SetOption (ENCODER_OPTION_FRAME_RATE) before frame N
    else if (arg_eq (a, "-dumpinfo")) infofile = next();
    else if (arg_eq (a, "-simulcast")) { low_w = std::atoi (next()); low_h = std::atoi (next()); if (nlow < 3) { lows[nlow][0] = low_w; lows[nlow][1] = low_h; ++nlow; } }
    else if (arg_eq (a, "-paramsets")) paramsets_at = std::atoi (next());       // EncodeParameterSets before frame N, output appended
    else if (arg_eq (a, "-quiet")) quiet = 1;
    else if (arg_eq (a, "-setbr")) { setbr_at = std::atoi (next()); setbr_val = std::atoi (next()); }
    else if (arg_eq (a, "-setres")) { setres_at = std::atoi (next()); setres_w = std::atoi (next()); setres_h = std::atoi (next()); }
    else if (arg_eq (a, "-setltr")) { setltr_at = std::atoi (next()); setltr_val = std::atoi (next()); }
    else if (arg_eq (a, "-ltrrecover")) ltrrecover_at = std::atoi (next());
    else if (arg_eq (a, "-ltrmarkfb")) ltrmarkfb_at = std::atoi (next());
    else if (arg_eq (a, "-tracelevel")) trace_level = std::atoi (next());      // ENCODER_OPTION_TRACE_LEVEL (WELS_LOG_INFO = 4): the encoder's own log on stderr
    else { std::fprintf (stderr, "unknown option %s\n", a); return 2; }
  }
  if (in.empty() || w <= 0 || h <= 0) { std::fprintf (stderr, "need -i -w -h\n"); return 2; }
  if (instance >= 0 && !out.empty()) out += "." + std::to_string (instance);

  ISVCEncoder* enc = NULL;
  if (WelsCreateSVCEncoder (&enc) || !enc) { std::fprintf (stderr, "WelsCreateSVCEncoder failed\n"); return 1; }
  int trace = trace_level >= 0 ? trace_level : quiet ? WELS_LOG_QUIET : WELS_LOG_ERROR;
  enc->SetOption (ENCODER_OPTION_TRACE_LEVEL, &trace);

  int ret;
  if (use_base) {            // the fixture used by test/api/BaseEncoderTest.cpp:8-24 (SEncParamBase)
    SEncParamBase b; std::memset (&b, 0, sizeof (b));
    b.iUsageType = (EUsageType)usage; b.iPicWidth = w; b.iPicHeight = h;
    b.iTargetBitrate = bitrate; b.iRCMode = (RC_MODES)rc; b.fMaxFrameRate = fps;
    ret = enc->Initialize (&b);
  } else if (use_ext) {     // the fixture test/api/BaseEncoderTest.cpp:25-71 uses when SEncParamBase cannot express the case (-denoise, -ltr, -cabac,
    SEncParamExt p; enc->GetDefaultParams (&p);      // -lossless, a slice mode): GetDefaultParams plus exactly those fields
    p.iUsageType = (EUsageType)usage; p.fMaxFrameRate = fps; p.iPicWidth = w; p.iPicHeight = h; p.iTargetBitrate = 5000000;
    p.bEnableDenoise = denoise != 0; p.iSpatialLayerNum = 1; p.bIsLosslessLink = lossless != 0;
    p.bEnableLongTermReference = ltr != 0; p.iEntropyCodingModeFlag = cabac ? 1 : 0;
    if (slcmd != SM_SINGLE_SLICE && slcmd != SM_SIZELIMITED_SLICE) p.iMultipleThreadIdc = 2;
    SSpatialLayerConfig& l = p.sSpatialLayers[0];
    l.iVideoWidth = w; l.iVideoHeight = h; l.fFrameRate = fps; l.iSpatialBitrate = p.iTargetBitrate;
    l.sSliceArgument.uiSliceMode = (SliceModeEnum)slcmd;
    if (slcmd == SM_SIZELIMITED_SLICE) { l.sSliceArgument.uiSliceSizeConstraint = 600; p.uiMaxNalSize = 1500; p.iMultipleThreadIdc = 4; p.bUseLoadBalancing = false; }
    if (slcmd == SM_FIXEDSLCNUM_SLICE) { l.sSliceArgument.uiSliceNum = 4; p.iMultipleThreadIdc = 4; p.bUseLoadBalancing = false; }
    if (cabac) l.uiProfileIdc = PRO_MAIN;
    if (threads != 1) p.iMultipleThreadIdc = (unsigned short)threads;      // (-threads N after -ext overrides the fixture's thread count)
    ret = enc->InitializeExt (&p);
  } else {
    SEncParamExt p; enc->GetDefaultParams (&p);
    p.iUsageType = (EUsageType)usage;
    p.iPicWidth = w; p.iPicHeight = h;
    p.iTargetBitrate = bitrate; p.iRCMode = (RC_MODES)rc; p.fMaxFrameRate = fps;
    p.iTemporalLayerNum = numtl; p.iSpatialLayerNum = 1;
    p.iComplexityMode = (ECOMPLEXITY_MODE)complexity;
    p.uiIntraPeriod = (unsigned)iper;
    p.eSpsPpsIdStrategy = (EParameterSetStrategy)spsid;
    p.iEntropyCodingModeFlag = cabac;
    p.bEnableFrameSkip = frameskip != 0;
    p.iMultipleThreadIdc = (unsigned short)threads;
    p.bUseLoadBalancing = loadbal != 0;
    p.iLoopFilterDisableIdc = deblock;
    p.iLoopFilterAlphaC0Offset = alpha; p.iLoopFilterBetaOffset = beta;
    p.bEnableFrameCroppingFlag = crop != 0;
    p.bEnableDenoise = denoise != 0;
    p.bEnableBackgroundDetection = bgd != 0;
    p.bEnableAdaptiveQuant = aq != 0;
    p.bEnableSceneChangeDetect = scene != 0;
    p.bEnableLongTermReference = ltr != 0;
    p.iMaxQp = 51; p.iMinQp = 0;
    SSpatialLayerConfig& l = p.sSpatialLayers[0];
    l.iVideoWidth = w; l.iVideoHeight = h; l.fFrameRate = fps;
    l.iSpatialBitrate = bitrate; l.iDLayerQp = qp;
    l.uiProfileIdc = (EProfileIdc)profile;       // default: layer2.cfg ProfileIdc 66
    l.sSliceArgument.uiSliceMode = (SliceModeEnum)slcmd;
    l.sSliceArgument.uiSliceNum = (unsigned)slcnum;
    if (slcmbnum > 0) for (int k = 0; k < MAX_SLICES_NUM_TMP; ++k) l.sSliceArgument.uiSliceMbNum[k] = (unsigned)slcmbnum;
    if (slcsize > 0) l.sSliceArgument.uiSliceSizeConstraint = (unsigned)slcsize;
    if (nalsize > 0) p.uiMaxNalSize = (unsigned)nalsize;
    if (nlow > 0) {                    // layers 0 .. nlow-1 = the lower resolutions (lowest first), layer nlow = the input resolution
      p.iSpatialLayerNum = nlow + 1; p.bSimulcastAVC = true;
      const SSpatialLayerConfig top = l;
      p.sSpatialLayers[nlow] = top;
      for (int k = 0; k < nlow; ++k) { p.sSpatialLayers[k] = top; p.sSpatialLayers[k].iVideoWidth = lows[k][0]; p.sSpatialLayers[k].iVideoHeight = lows[k][1]; }
      p.iTargetBitrate = bitrate * (nlow + 1);       // every layer gets `bitrate`; the total must cover their sum
    }
    ret = enc->InitializeExt (&p);
  }
  if (ret) { std::fprintf (stderr, "Initialize failed: %d\n", ret); return 1; }

  FILE* fi = std::fopen (in.c_str(), "rb");
  if (!fi) { std::fprintf (stderr, "cannot open %s\n", in.c_str()); return 1; }
  FILE* fo = out.empty() ? NULL : std::fopen (out.c_str(), "wb");
  size_t fsz = (size_t)w * h * 3 / 2;
  std::vector<unsigned char> buf (fsz);
  int idr_pics = 0, frame_num = 0;     // what a receiver would know: IDR pictures so far, frame_num of the picture before (reference pictures only: one temporal layer assumed)
  SSourcePicture pic; std::memset (&pic, 0, sizeof (pic));
  pic.iColorFormat = videoFormatI420; pic.iPicWidth = w; pic.iPicHeight = h;
  pic.iStride[0] = w; pic.iStride[1] = pic.iStride[2] = w >> 1;
  pic.pData[0] = buf.data(); pic.pData[1] = buf.data() + (size_t)w * h; pic.pData[2] = pic.pData[1] + (size_t) (w >> 1) * (h >> 1);
  SFrameBSInfo info;
  long long total = 0; int n = 0; double secs = 0.0;
  while (frames < 0 || n < frames) {
    if (n == setres_at) {
      SEncParamExt q; std::memset (&q, 0, sizeof (q));
      if (enc->GetOption (ENCODER_OPTION_SVC_ENCODE_PARAM_EXT, &q)) { std::fprintf (stderr, "GetOption (SVC_ENCODE_PARAM_EXT) failed\n"); return 1; }
      w = setres_w; h = setres_h;
      q.iPicWidth = w; q.iPicHeight = h; q.sSpatialLayers[q.iSpatialLayerNum - 1].iVideoWidth = w; q.sSpatialLayers[q.iSpatialLayerNum - 1].iVideoHeight = h;
      if (enc->SetOption (ENCODER_OPTION_SVC_ENCODE_PARAM_EXT, &q)) { std::fprintf (stderr, "SetOption (SVC_ENCODE_PARAM_EXT) failed\n"); return 1; }
      fsz = (size_t)w * h * 3 / 2; buf.resize (fsz);       // the rest of the input file holds frames of the new size
      pic.iPicWidth = w; pic.iPicHeight = h; pic.iStride[0] = w; pic.iStride[1] = pic.iStride[2] = w >> 1;
      pic.pData[0] = buf.data(); pic.pData[1] = buf.data() + (size_t)w * h; pic.pData[2] = pic.pData[1] + (size_t) (w >> 1) * (h >> 1);
    }
    if (std::fread (buf.data(), 1, fsz, fi) != fsz) break;
    std::memset (&info, 0, sizeof (info));
    pic.uiTimeStamp = (long long) (n * (1000.0 / fps) + 0.5);
    if (n == forceidr) enc->ForceIntraFrame (true);
    if (n == setidr_at) enc->SetOption (ENCODER_OPTION_IDR_INTERVAL, &setidr_val);
    if (n == setcplx_at) enc->SetOption (ENCODER_OPTION_COMPLEXITY, &setcplx_val);
    if (n == setfps_at) enc->SetOption (ENCODER_OPTION_FRAME_RATE, &setfps_val);
    if (n == setbr_at) { SBitrateInfo b; std::memset (&b, 0, sizeof (b)); b.iLayer = SPATIAL_LAYER_ALL; b.iBitrate = setbr_val; if (enc->SetOption (ENCODER_OPTION_BITRATE, &b)) { std::fprintf (stderr, "SetOption (BITRATE) failed\n"); return 1; } }
    if (n == setltr_at) { SLTRConfig c; std::memset (&c, 0, sizeof (c)); c.bEnableLongTermReference = setltr_val != 0; c.iLTRRefNum = 2; enc->SetOption (ENCODER_OPTION_LTR, &c); }
    if (n == ltrrecover_at) {
      SLTRRecoverRequest r; std::memset (&r, 0, sizeof (r));
      r.uiFeedbackType = LTR_RECOVERY_REQUEST; r.uiIDRPicId = (unsigned)(idr_pics > 0 ? idr_pics - 1 : 0); r.iLastCorrectFrameNum = frame_num > 1 ? frame_num - 2 : 0; r.iCurrentFrameNum = frame_num > 0 ? frame_num - 1 : 0;
      enc->SetOption (ENCODER_LTR_RECOVERY_REQUEST, &r);
    }
    if (n == ltrmarkfb_at) {
      SLTRMarkingFeedback f; std::memset (&f, 0, sizeof (f));
      f.uiFeedbackType = LTR_MARKING_SUCCESS; f.uiIDRPicId = (unsigned)(idr_pics > 0 ? idr_pics - 1 : 0); f.iLTRFrameNum = frame_num > 0 ? frame_num - 1 : 0;
      enc->SetOption (ENCODER_LTR_MARKING_FEEDBACK, &f);
    }
    if (n == paramsets_at) {
      SFrameBSInfo ps; std::memset (&ps, 0, sizeof (ps));
      if (enc->EncodeParameterSets (&ps)) { std::fprintf (stderr, "EncodeParameterSets failed\n"); return 1; }
      for (int li = 0; li < ps.iLayerNum; ++li) {
        const SLayerBSInfo& L = ps.sLayerInfo[li];
        int sz = 0; for (int k = 0; k < L.iNalCount; ++k) sz += L.pNalLengthInByte[k];
        if (fo) std::fwrite (L.pBsBuf, 1, sz, fo);
        total += sz;
      }
    }
    auto t0 = std::chrono::steady_clock::now();
    ret = enc->EncodeFrame (&pic, &info);
    auto t1 = std::chrono::steady_clock::now();
    secs += std::chrono::duration<double> (t1 - t0).count();
    if (ret) { std::fprintf (stderr, "EncodeFrame failed: %d\n", ret); return 1; }
    if (info.eFrameType == videoFrameTypeIDR) { ++idr_pics; frame_num = 1; }
    else if (info.eFrameType == videoFrameTypeI || info.eFrameType == videoFrameTypeP) ++frame_num;
    if (!infofile.empty()) {
      FILE* fm = std::fopen (infofile.c_str(), n == 0 ? "w" : "a");
      std::fprintf (fm, "frame %d type %d layers %d size %d ts %lld\n", n, (int)info.eFrameType, info.iLayerNum, info.iFrameSizeInBytes, (long long)info.uiTimeStamp);
      for (int li = 0; li < info.iLayerNum; ++li) {
        const SLayerBSInfo& L = info.sLayerInfo[li];
        std::fprintf (fm, "  layer %d ltype %d ftype %d tid %d sid %d qid %d subseq %d nals", li, (int)L.uiLayerType, (int)L.eFrameType, (int)L.uiTemporalId, (int)L.uiSpatialId, (int)L.uiQualityId, L.iSubSeqId);
        for (int k = 0; k < L.iNalCount; ++k) std::fprintf (fm, " %d", L.pNalLengthInByte[k]);
        std::fprintf (fm, "\n");
      }
      std::fclose (fm);
    }
    if (info.eFrameType != videoFrameTypeSkip) {
      for (int li = 0; li < info.iLayerNum; ++li) {
        const SLayerBSInfo& L = info.sLayerInfo[li];
        int sz = 0; for (int k = 0; k < L.iNalCount; ++k) sz += L.pNalLengthInByte[k];
        if (fo) std::fwrite (L.pBsBuf, 1, sz, fo);
        total += sz;
        if (low_w > 0 && !quiet) std::fprintf (stderr, "frame %d layer %d: type %d spatial %d nals %d bytes %d\n", n, li, (int)L.uiLayerType, (int)L.uiSpatialId, L.iNalCount, sz);
      }
    }
    ++n;
  }
  if (fo) std::fclose (fo);
  std::fclose (fi);
  enc->Uninitialize();
  WelsDestroySVCEncoder (enc);
  std::printf ("frames=%d bytes=%lld enc_seconds=%.6f fps=%.3f\n", n, total, secs, secs > 0 ? n / secs : 0.0);
  return 0;
}

// -parallel N (first argument pair): N encoder instances on N threads of this one process, all with the remaining options --
// what a media server hosting several sessions does.  Prints every instance's line and the aggregate.
int main (int argc, char** argv) {
  if (argc > 2 && arg_eq (argv[1], "-parallel")) {
    const int n = std::atoi (argv[2]);
    std::vector<char*> args;
    args.push_back (argv[0]);
    for (int i = 3; i < argc; ++i) args.push_back (argv[i]);
    std::vector<int> rcs ((size_t)n, 0);
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < n; ++k) th.emplace_back ([&, k] { rcs[(size_t)k] = run ((int)args.size(), args.data(), k); });
    for (auto& t : th) t.join();
    const double wall = std::chrono::duration<double> (std::chrono::steady_clock::now() - t0).count();
    int bad = 0;
    for (int r : rcs) bad += r != 0;
    std::printf ("parallel=%d wall_seconds=%.6f failed=%d\n", n, wall, bad);
    return bad ? 1 : 0;
  }
  return run (argc, argv, -1);
}

