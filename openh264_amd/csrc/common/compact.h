// compact.h -- the macroblock records as they cross PCIe (shared by the device kernel, the CPU test build and the host).
//
// A WhMbRecord is 960 bytes, of which a P picture needs a fraction: a skipped MB carries no syntax at all, most coded MBs
// have a handful of non-zero 4x4 blocks.  At the device's rate (thousands of 1080p pictures per second) the full records
// are more than a PCIe Gen5 x16 link moves, so the records of a picture are packed on the device before they are copied:
//
//   off[mb]            uint32 byte offset of MB `mb` in the packed stream, off[num_mb] = total size
//   P_Skip MB          16 bytes: record bytes 0..7 (type, cbp, QPs, modes), cost (int32), bgd_skip, 3 x 0
//   any other MB       4-byte block mask + the 144-byte side info + 32 bytes per block whose mask bit is set
//                      bits 0..15  luma[b] (luma4x4BlkIdx b), bit 16 luma_dc, bits 17..24 chroma_ac[k], bit 25 chroma_dc (16 bytes)
//                      A block is sent when it has a non-zero level AND the coded_block_pattern lets the entropy coder read it.
//
// The host expands a picture back into WhMbRecord[] (zero-filled where nothing was sent) and entropy-codes from that: the
// writer sees exactly what it would have seen from the full records.
#pragma once
#include <stdint.h>
#include <string.h>
#include "wh_types.h"
#if defined(__HIPCC__)
#define WH_CP_FN static __host__ __device__ inline
#else
#define WH_CP_FN static inline
#endif

#define WH_COMPACT_SIDE 144u           /* offsetof (WhMbRecord, luma) */
#define WH_COMPACT_SKIP_BYTES 16u
#define WH_COMPACT_MAX_BYTES (4u + 960u)

// which blocks of a record the entropy coder can read at all, from mb_type and cbp
WH_CP_FN uint32_t wh_compact_allowed (int mb_type, int cbp) {
  uint32_t m = 0;
  if (mb_type == WH_MB_I16x16) { m |= 1u << 16; if (cbp & 15) m |= 0xffffu; }
  else for (int k = 0; k < 4; ++k) if ((cbp >> k) & 1) m |= 0xfu << (4 * k);
  if ((cbp >> 4) >= 1) m |= 1u << 25;
  if ((cbp >> 4) == 2) m |= 0xffu << 17;
  return m;
}
WH_CP_FN uint32_t wh_compact_size (int mb_type, uint32_t mask) {
  if (mb_type == WH_MB_PSKIP) return WH_COMPACT_SKIP_BYTES;
  return 4u + WH_COMPACT_SIDE + 32u * (uint32_t)__builtin_popcount (mask & 0x1ffffffu) + ((mask >> 25) & 1u) * 16u;
}

// host side: one packed MB -> a full record
static inline void wh_compact_expand (const uint8_t* p, uint32_t size, WhMbRecord* R) {
  memset (R, 0, sizeof (*R));
  if (size == WH_COMPACT_SKIP_BYTES) {
    memcpy (R, p, 8);
    memcpy (&R->cost, p + 8, 4);
    R->bgd_skip = p[12];
    return;
  }
  uint32_t mask;
  memcpy (&mask, p, 4);
  memcpy (R, p + 4, WH_COMPACT_SIDE);
  const uint8_t* q = p + 4 + WH_COMPACT_SIDE;
  for (int b = 0; b < 16; ++b) if ((mask >> b) & 1u) { memcpy (R->luma[b], q, 32); q += 32; }
  if ((mask >> 16) & 1u) { memcpy (R->luma_dc, q, 32); q += 32; }
  for (int k = 0; k < 8; ++k) if ((mask >> (17 + k)) & 1u) { memcpy (R->chroma_ac[k], q, 32); q += 32; }
  if ((mask >> 25) & 1u) memcpy (R->chroma_dc, q, 16);
}
// reference packer (CPU test build; also documents what the device kernel does)
static inline uint32_t wh_compact_pack (const WhMbRecord* R, uint8_t* out) {
  if (R->mb_type == WH_MB_PSKIP) {
    memset (out, 0, WH_COMPACT_SKIP_BYTES);
    memcpy (out, R, 8);
    memcpy (out + 8, &R->cost, 4);
    out[12] = R->bgd_skip;
    return WH_COMPACT_SKIP_BYTES;
  }
  const uint32_t allowed = wh_compact_allowed (R->mb_type, R->cbp);
  uint32_t mask = 0;
  const int16_t* c = &R->luma[0][0];
  for (int b = 0; b < 25; ++b) { bool nz = false; for (int k = 0; k < 16; ++k) nz = nz || c[b * 16 + k] != 0; if (nz) mask |= 1u << b; }
  { bool nz = false; for (int k = 0; k < 8; ++k) nz = nz || c[400 + k] != 0; if (nz) mask |= 1u << 25; }
  mask &= allowed;
  memcpy (out, &mask, 4);
  memcpy (out + 4, R, WH_COMPACT_SIDE);
  uint8_t* q = out + 4 + WH_COMPACT_SIDE;
  for (int b = 0; b < 25; ++b) if ((mask >> b) & 1u) { memcpy (q, c + b * 16, 32); q += 32; }
  if ((mask >> 25) & 1u) { memcpy (q, c + 400, 16); q += 16; }
  return (uint32_t) (q - out);
}
