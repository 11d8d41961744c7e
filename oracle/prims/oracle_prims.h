/* oracle_prims.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's per-macroblock integer primitives (its C fallback, the
 * `_c` functions).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * into this library; the product never links or loads it.  Every function cites the reference
 * file:line it follows (paths relative to the cisco/openh264 tree).  The restatement is pinned
 * against the real reference (oracle/_ref/libref_prims.so exports the same functions with the
 * prefix ref_ instead of orc_) by tests/test_oracle_prims.py in the build container.
 */
#ifndef ORACLE_PRIMS_H_
#define ORACLE_PRIMS_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* block size index as in codec/encoder/core/inc/wels_const.h:139-148 */
enum { ORC_BLOCK_16x16 = 0, ORC_BLOCK_16x8, ORC_BLOCK_8x16, ORC_BLOCK_8x8, ORC_BLOCK_4x4, ORC_BLOCK_8x4, ORC_BLOCK_4x8 };

/* SSampleDealingFunc: pfSampleSad / pfSample4Sad / pfSampleSatd */
int32_t orc_sad (int blk, const uint8_t* a, int32_t sa, const uint8_t* b, int32_t sb);
void    orc_sad_four (int blk, const uint8_t* a, int32_t sa, const uint8_t* b, int32_t sb, int32_t* out4);
int32_t orc_satd (int blk, const uint8_t* a, int32_t sa, const uint8_t* b, int32_t sb);

/* pfDctT4 / pfQuantization* / pfScan4x4* / pfDequantization* / pfIDctT4 ... */
void    orc_dct4x4 (int16_t* dct, const uint8_t* pix1, int32_t s1, const uint8_t* pix2, int32_t s2);
void    orc_hadamard_t4_dc (int16_t* luma_dc, const int16_t* dct256);
void    orc_quant4x4 (int16_t* dct, int qp, int intra);
int32_t orc_quant4x4_max (int16_t* dct, int qp, int intra);            /* returns the block's max |level| */
void    orc_quant4x4_dc (int16_t* dct, int16_t ff, int16_t mf);
void    orc_quant_rows (int qp, int intra, int16_t* ff8, int16_t* mf8);         /* the table rows the encoder passes to pfQuantization* */
int32_t orc_hadamard_quant2x2 (int16_t* rs, int16_t ff, int16_t mf, int16_t* dct4, int16_t* block4);
int32_t orc_hadamard_quant2x2_skip (const int16_t* rs, int16_t ff, int16_t mf);
void    orc_scan4x4_dcac (int16_t* level, const int16_t* dct);
void    orc_scan4x4_ac (int16_t* level, const int16_t* dct);
int32_t orc_single_ctr4x4 (const int16_t* level);
int32_t orc_nonzero_count (const int16_t* level);
void    orc_dequant4x4 (int16_t* res, int qp);
void    orc_dequant_ihadamard4x4 (int16_t* res, int qp);              /* luma DC, any qp (both branches) */
void    orc_dequant_ihadamard2x2_dc (int16_t* dct4, int qp);
void    orc_idct4x4_rec (uint8_t* rec, int32_t rs, const uint8_t* pred, int32_t ps, const int16_t* dct);

/* intra predictors: mode numbering of the reference (I4_PRED_*, I16_PRED_*, C_PRED_*); ref points at
 * the block's top-left sample inside a picture with `stride`; pred is packed 4x4 / 16x16 / 8x8 */
void    orc_pred_i4x4 (int mode, uint8_t* pred, const uint8_t* ref, int32_t stride);
void    orc_pred_i16x16 (int mode, uint8_t* pred, const uint8_t* ref, int32_t stride);
void    orc_pred_chroma (int mode, uint8_t* pred, const uint8_t* ref, int32_t stride);

/* SMcFunc: pMcLumaFunc / pMcChromaFunc */
void    orc_mc_luma (const uint8_t* src, int32_t ss, uint8_t* dst, int32_t ds, int mvx, int mvy, int w, int h);
void    orc_mc_chroma (const uint8_t* src, int32_t ss, uint8_t* dst, int32_t ds, int mvx, int mvy, int w, int h);

/* DeblockingFunc edge filters; horizontal != 0 filters a vertical edge (samples step by 1) */
void    orc_deblock_luma_lt4 (uint8_t* pix, int32_t stride, int horizontal, int alpha, int beta, const int8_t* tc4);
void    orc_deblock_luma_eq4 (uint8_t* pix, int32_t stride, int horizontal, int alpha, int beta);
void    orc_deblock_chroma_lt4 (uint8_t* pix, int32_t stride, int horizontal, int alpha, int beta, const int8_t* tc4);
void    orc_deblock_chroma_eq4 (uint8_t* pix, int32_t stride, int horizontal, int alpha, int beta);

/* VAA (codec/processing): four 8x8 SADs of one MB against the previous source picture */
void    orc_vaa_sad8x8 (const uint8_t* cur, const uint8_t* ref, int32_t stride, int32_t* sad4);

/* batch.c: loops of the functions above over arrays of cases (one call per draw) */
void    orc_sad_batch (int blk, int n, const uint8_t* p1, int32_t s1, const int32_t* o1, const uint8_t* p2, int32_t s2, const int32_t* o2, int32_t* out);
void    orc_satd_batch (int blk, int n, const uint8_t* p1, int32_t s1, const int32_t* o1, const uint8_t* p2, int32_t s2, const int32_t* o2, int32_t* out);
void    orc_sad_four_batch (int blk, int n, const uint8_t* p1, int32_t s1, const int32_t* o1, const uint8_t* p2, int32_t s2, const int32_t* o2, int32_t* out4);
void    orc_dct4x4_batch (int n, const uint8_t* p1, int32_t s1, const int32_t* o1, const uint8_t* p2, int32_t s2, const int32_t* o2, int16_t* out);
void    orc_quant_scan_batch (int n, int16_t* io, const uint8_t* qp, int intra, int16_t* mx, int16_t* zz, int16_t* za, int32_t* ctr, int32_t* nz);
void    orc_dequant_idct_rec_batch (int n, const int16_t* lev, const uint8_t* qp, const uint8_t* pred, uint8_t* rec, int16_t* deq);

#ifdef __cplusplus
}
#endif
#endif
