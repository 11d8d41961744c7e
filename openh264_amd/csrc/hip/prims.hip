// prims.hip -- layer (3) of the C ABI: the leaf primitives of SWelsFuncPtrList / SMcFunc, batched over
// arrays of blocks.  Same per-block semantics as the reference's function-pointer entries
// (codec/encoder/core/inc/wels_func_ptr_def.h:58-296, codec/common/inc/mc.h:40-53); each kernel calls
// the very device functions the fused macroblock kernels use (kernels/prims.h, intra_mb.h, inter_mb.h,
// deblock_mb.h), so the parity tests of this layer pin the arithmetic of the hot path itself.
// Buffers are HOST pointers; the library stages them through HBM (this layer exists for parity
// testing and integration bring-up -- throughput comes from the frame-level kernels).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <initializer_list>
#include "../../../include/welship.h"
#include "../kernels/frame_kernels.h"
#include "../kernels/inter_mb.h"
#include "../kernels/mc_px.h"
#include "../kernels/deblock_mb.h"

namespace {

struct Dev {          // RAII device copy of a host array
  void* p = nullptr; size_t n = 0;
  Dev (const void* h, size_t bytes) : n (bytes) { if (hipMalloc (&p, bytes ? bytes : 1) != hipSuccess) p = nullptr; else if (h && bytes) (void)hipMemcpy (p, h, bytes, hipMemcpyHostToDevice); }
  ~Dev() { if (p) (void)hipFree (p); }
  template <class T> T* as() const { return (T*)p; }
  void back (void* h) const { (void)hipMemcpy (h, p, n, hipMemcpyDeviceToHost); }
};
// every staging buffer of a call was allocated (hipMalloc can fail: the caller gets cmMallocMemeError, not a fault)
bool dev_ok (std::initializer_list<const Dev*> l) { for (const Dev* d : l) if (!d->p) return false; return true; }
#define DEV_OK(...) do { if (!dev_ok ({__VA_ARGS__})) return WELSHIP_ERR_MEMORY; } while (0)
bool have_gpu() { int c = 0; return hipGetDeviceCount (&c) == hipSuccess && c > 0; }
#define NEED_GPU() do { if (!have_gpu()) return WELSHIP_ERR_NO_DEVICE; } while (0)
inline int grid (int n) { return (n + 255) / 256; }

#include "prims_kernels.h"
}  // namespace

extern "C" {

static int sad_like (int mode, int blk, int n, const uint8_t* p1, size_t b1, int s1, const int32_t* o1, const uint8_t* p2, size_t b2, int s2, const int32_t* o2, int32_t* out) {
  NEED_GPU();
  if (blk < 0 || blk > 6 || n <= 0) return WELSHIP_ERR_INIT_PARA;
  Dev d1 (p1, b1), d2 (p2, b2), do1 (o1, n * 4), do2 (o2, n * 4), dout (nullptr, (size_t)n * 4 * (mode == 2 ? 4 : 1));
  DEV_OK (&d1, &d2, &do1, &do2, &dout);
  if (blk <= 3) hipLaunchKernelGGL (k_sad_wave, dim3 (n), dim3 (64), 0, 0, blk, d1.as<uint8_t>(), s1, do1.as<int>(), d2.as<uint8_t>(), b2, s2, do2.as<int>(), dout.as<int>(), mode);
  else hipLaunchKernelGGL (k_sad, dim3 (grid (n)), dim3 (256), 0, 0, blk, n, d1.as<uint8_t>(), s1, do1.as<int>(), d2.as<uint8_t>(), s2, do2.as<int>(), dout.as<int>(), mode);
  if (hipDeviceSynchronize() != hipSuccess) return WELSHIP_ERR_UNKNOWN;
  dout.back (out);
  return WELSHIP_OK;
}
int WelsHipPrimSampleSad (int iBlock, int n, const uint8_t* p1, size_t b1, int32_t s1, const int32_t* o1, const uint8_t* p2, size_t b2, int32_t s2, const int32_t* o2, int32_t* pSad) { return sad_like (0, iBlock, n, p1, b1, s1, o1, p2, b2, s2, o2, pSad); }
int WelsHipPrimSampleSatd (int iBlock, int n, const uint8_t* p1, size_t b1, int32_t s1, const int32_t* o1, const uint8_t* p2, size_t b2, int32_t s2, const int32_t* o2, int32_t* pSatd) { return sad_like (1, iBlock, n, p1, b1, s1, o1, p2, b2, s2, o2, pSatd); }
int WelsHipPrimSample4Sad (int iBlock, int n, const uint8_t* p1, size_t b1, int32_t s1, const int32_t* o1, const uint8_t* p2, size_t b2, int32_t s2, const int32_t* o2, int32_t* pSad4) { return sad_like (2, iBlock, n, p1, b1, s1, o1, p2, b2, s2, o2, pSad4); }

int WelsHipPrimDctT4 (int n, const uint8_t* p1, size_t b1, int32_t s1, const int32_t* o1, const uint8_t* p2, size_t b2, int32_t s2, const int32_t* o2, int16_t* pDct) {
  NEED_GPU();
  Dev d1 (p1, b1), d2 (p2, b2), do1 (o1, n * 4), do2 (o2, n * 4), dout (nullptr, (size_t)n * 32);
  DEV_OK (&d1, &d2, &do1, &do2, &dout);
  hipLaunchKernelGGL (k_dct, dim3 (grid (n)), dim3 (256), 0, 0, n, d1.as<uint8_t>(), s1, do1.as<int>(), d2.as<uint8_t>(), s2, do2.as<int>(), dout.as<int16_t>());
  if (hipDeviceSynchronize() != hipSuccess) return WELSHIP_ERR_UNKNOWN;
  dout.back (pDct);
  return WELSHIP_OK;
}
int WelsHipPrimQuant4x4 (int n, int16_t* pDctInOut, const uint8_t* pQp, int bIntra, int16_t* pMax, int16_t* pScanDcAc, int16_t* pScanAc, int32_t* pSingleCtr, int32_t* pNzc) {
  NEED_GPU();
  Dev dio (pDctInOut, (size_t)n * 32), dqp (pQp, n), dmx (nullptr, (size_t)n * 2), dzz (nullptr, (size_t)n * 32), dza (nullptr, (size_t)n * 32), dct (nullptr, (size_t)n * 4), dnz (nullptr, (size_t)n * 4);
  DEV_OK (&dio, &dqp, &dmx, &dzz, &dza, &dct, &dnz);
  hipLaunchKernelGGL (k_quant, dim3 (grid (n)), dim3 (256), 0, 0, n, dio.as<int16_t>(), dqp.as<uint8_t>(), bIntra, dmx.as<int16_t>(), dzz.as<int16_t>(), dza.as<int16_t>(), dct.as<int>(), dnz.as<int>());
  if (hipDeviceSynchronize() != hipSuccess) return WELSHIP_ERR_UNKNOWN;
  dio.back (pDctInOut); dmx.back (pMax); dzz.back (pScanDcAc); dza.back (pScanAc); dct.back (pSingleCtr); dnz.back (pNzc);
  return WELSHIP_OK;
}
int WelsHipPrimDequantIDctRec (int n, const int16_t* pLevelsRaster, const uint8_t* pQp, const uint8_t* pPred, uint8_t* pRec, int16_t* pDequant) {
  NEED_GPU();
  Dev dc (pLevelsRaster, (size_t)n * 32), dqp (pQp, n), dp (pPred, (size_t)n * 16), dr (nullptr, (size_t)n * 16), dd (nullptr, (size_t)n * 32);
  DEV_OK (&dc, &dqp, &dp, &dr, &dd);
  hipLaunchKernelGGL (k_dequant_idct, dim3 (grid (n)), dim3 (256), 0, 0, n, dc.as<int16_t>(), dqp.as<uint8_t>(), dp.as<uint8_t>(), dr.as<uint8_t>(), dd.as<int16_t>());
  if (hipDeviceSynchronize() != hipSuccess) return WELSHIP_ERR_UNKNOWN;
  dr.back (pRec); dd.back (pDequant);
  return WELSHIP_OK;
}
int WelsHipPrimIntraPred4x4 (int n, const uint8_t* pPlane, size_t bytes, int32_t iStride, const int32_t* pOff, const uint8_t* pMode, const uint8_t* pAvail, uint8_t* pPred) {
  NEED_GPU();
  Dev dp (pPlane, bytes), dof (pOff, n * 4), dm (pMode, n), da (pAvail, n), dout (nullptr, (size_t)n * 16);
  DEV_OK (&dp, &dof, &dm, &da, &dout);
  hipLaunchKernelGGL (k_pred4, dim3 (grid (n)), dim3 (256), 0, 0, n, dp.as<uint8_t>(), iStride, dof.as<int>(), dm.as<uint8_t>(), da.as<uint8_t>(), dout.as<uint8_t>());
  if (hipDeviceSynchronize() != hipSuccess) return WELSHIP_ERR_UNKNOWN;
  dout.back (pPred);
  return WELSHIP_OK;
}
int WelsHipPrimIntraPredMb (int n, const uint8_t* pPlaneY, size_t bytesY, int32_t iStrideY, const int32_t* pOffY, const uint8_t* pPlaneC, size_t bytesC, int32_t iStrideC, const int32_t* pOffC,
                            const uint8_t* pMode16, const uint8_t* pModeChroma, uint8_t* pPred16, uint8_t* pPredChroma) {
  NEED_GPU();
  Dev dy (pPlaneY, bytesY), doy (pOffY, n * 4), dc (pPlaneC, bytesC), doc (pOffC, n * 4), dm (pMode16, n), dmc (pModeChroma, n), o16 (nullptr, (size_t)n * 256), oc (nullptr, (size_t)n * 128);
  DEV_OK (&dy, &doy, &dc, &doc, &dm, &dmc, &o16, &oc);
  hipLaunchKernelGGL (k_pred_mb, dim3 (n), dim3 (64), 0, 0, dy.as<uint8_t>(), iStrideY, doy.as<int>(), dc.as<uint8_t>(), iStrideC, doc.as<int>(), dm.as<uint8_t>(), dmc.as<uint8_t>(), o16.as<uint8_t>(), oc.as<uint8_t>());
  if (hipDeviceSynchronize() != hipSuccess) return WELSHIP_ERR_UNKNOWN;
  o16.back (pPred16); oc.back (pPredChroma);
  return WELSHIP_OK;
}
int WelsHipPrimMc (int n, const uint8_t* pPlane, size_t bytes, int32_t iStride, const int32_t* pOff, const int16_t* pMv, int iWidth, int iHeight, int bChroma, uint8_t* pDst) {
  NEED_GPU();
  Dev dp (pPlane, bytes), dof (pOff, n * 4), dmv (pMv, (size_t)n * 4), dout (nullptr, (size_t)n * iWidth * iHeight);
  DEV_OK (&dp, &dof, &dmv, &dout);
  if (!bChroma && (iWidth == 16 || iWidth == 8) && (iHeight == 16 || iHeight == 8))
    hipLaunchKernelGGL (k_mc_wave, dim3 (n), dim3 (64), 0, 0, dp.as<uint8_t>(), bytes, iStride, dof.as<int>(), dmv.as<int16_t>(), iWidth, iHeight, dout.as<uint8_t>());
  else
    hipLaunchKernelGGL (k_mc, dim3 (grid (n)), dim3 (256), 0, 0, n, dp.as<uint8_t>(), iStride, dof.as<int>(), dmv.as<int16_t>(), iWidth, iHeight, bChroma, dout.as<uint8_t>());
  if (hipDeviceSynchronize() != hipSuccess) return WELSHIP_ERR_UNKNOWN;
  dout.back (pDst);
  return WELSHIP_OK;
}
int WelsHipPrimDeblockEdges (int nEdges, uint8_t* pPlaneInOut, size_t bytes, int32_t iStride, const int32_t* pOff, int bVerticalEdge, int bChroma, const uint8_t* pBs4, const uint8_t* pIndexA) {
  NEED_GPU();
  Dev dp (pPlaneInOut, bytes), dof (pOff, nEdges * 4), dbs (pBs4, (size_t)nEdges * 4), dia (pIndexA, nEdges);
  DEV_OK (&dp, &dof, &dbs, &dia);
  const int lines = nEdges * (bChroma ? 8 : 16);
  hipLaunchKernelGGL (k_deblock, dim3 (grid (lines)), dim3 (256), 0, 0, nEdges, dp.as<uint8_t>(), iStride, dof.as<int>(), bVerticalEdge, bChroma, dbs.as<uint8_t>(), dia.as<uint8_t>());
  if (hipDeviceSynchronize() != hipSuccess) return WELSHIP_ERR_UNKNOWN;
  dp.back (pPlaneInOut);
  return WELSHIP_OK;
}
int WelsHipPrimVaaSad8x8 (int nMb, const uint8_t* pCur, const uint8_t* pRef, size_t bytes, int32_t iStride, const int32_t* pOff, int32_t* pSad8x8) {
  NEED_GPU();
  Dev dc (pCur, bytes), dr (pRef, bytes), dof (pOff, nMb * 4), dout (nullptr, (size_t)nMb * 16);
  DEV_OK (&dc, &dr, &dof, &dout);
  hipLaunchKernelGGL (k_vaa, dim3 (grid (nMb * 4)), dim3 (256), 0, 0, nMb, dc.as<uint8_t>(), dr.as<uint8_t>(), iStride, dof.as<int>(), dout.as<int>());
  if (hipDeviceSynchronize() != hipSuccess) return WELSHIP_ERR_UNKNOWN;
  dout.back (pSad8x8);
  return WELSHIP_OK;
}

}  // extern "C"

// layer (3c): the same kernels behind the reference's own function-pointer typedefs (include/welship_leaf.h)
#define WH_PRIMS_TU 1
#include "leaf.hip"
