/* oracle_prims.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's per-macroblock integer primitives (its C fallback, the
 * `_c` functions).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * into this library; the product never links or loads it.  Every function cites the reference
 * file:line it follows (paths relative to the cisco/openh264 tree).  The restatement is pinned
 * against the real reference (oracle/_ref/libref_prims.so, same signatures) by
 * tests/test_oracle_prims.py in the build container.
 */
#ifndef ORACLE_PRIMS_H_
#define ORACLE_PRIMS_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* block size index as in codec/encoder/core/inc/wels_const.h:139-148 */
enum { ORC_BLOCK_16x16 = 0, ORC_BLOCK_16x8, ORC_BLOCK_8x16, ORC_BLOCK_8x8, ORC_BLOCK_4x4, ORC_BLOCK_8x4, ORC_BLOCK_4x8 };

int32_t orc_sad (int blk, const uint8_t* a, int32_t sa, const uint8_t* b, int32_t sb);
void    orc_sad_four (int blk, const uint8_t* a, int32_t sa, const uint8_t* b, int32_t sb, int32_t* out4);
int32_t orc_satd (int blk, const uint8_t* a, int32_t sa, const uint8_t* b, int32_t sb);

#ifdef __cplusplus
}
#endif
#endif
