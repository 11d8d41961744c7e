// bitwriter.h -- RBSP bit writer, Exp-Golomb codes and Annex-B NAL encapsulation (host side).
//
// Syntax elements follow ITU-T H.264 7.2 / 9.1; behaviourally equivalent to the reference's
// codec/common/inc/golomb_common.h:102-163 and codec/encoder/core/src/nal_encap.cpp:118-185.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <vector>

namespace wh {

// Bits are collected MSB-first in a 64-bit accumulator and leave it four bytes at a time, straight into the vector's
// storage (grown geometrically, trimmed by finish()/trailing()); the vector's size is only meaningful after that.
class BitWriter {
 public:
  explicit BitWriter (std::vector<uint8_t>* out) : out_ (out), pos_ (out->size()) {}
  ~BitWriter () { finish(); }
  void put (int n, uint32_t v) {                 // n in [0,32], MSB first; bits of v above n are ignored
    if (n <= 0) return;
    acc_ = (acc_ << n) | (uint64_t) (n == 32 ? v : (v & ((1u << n) - 1u)));
    nbits_ += n;
    if (nbits_ >= 32) {
      if (pos_ + 4 > out_->size()) out_->resize (out_->size() * 2 + 4096);
      const uint32_t w = (uint32_t) (acc_ >> (nbits_ - 32));
      uint8_t* d = out_->data() + pos_;
      d[0] = (uint8_t) (w >> 24); d[1] = (uint8_t) (w >> 16); d[2] = (uint8_t) (w >> 8); d[3] = (uint8_t)w;
      pos_ += 4;
      nbits_ -= 32;
    }
  }
  void bit (int b) { put (1, b ? 1u : 0u); }
  void ue (uint32_t v) {
    const uint32_t x = v + 1;
    const int len = 31 - __builtin_clz (x);      // floor(log2(x)), x >= 1
    if (2 * len + 1 <= 32) put (2 * len + 1, x);
    else { put (len, 0); put (len + 1, x); }
  }
  void se (int32_t v) { ue (v > 0 ? (uint32_t) (2 * v - 1) : (uint32_t) (-2 * v)); }
  void te (int max_minus1, uint32_t v) { if (max_minus1 == 1) bit (!v); else ue (v); }
  void trailing () { bit (1); if (nbits_ & 7) put (8 - (nbits_ & 7), 0); finish(); }
  size_t bits () const { return pos_ * 8 + (size_t)nbits_; }
  // Flush whole bytes and give the vector its final size (call at a byte boundary; trailing() does).
  void finish () {
    while (nbits_ >= 8) {
      if (pos_ + 1 > out_->size()) out_->resize (out_->size() * 2 + 4096);
      (*out_)[pos_++] = (uint8_t) (acc_ >> (nbits_ - 8));
      nbits_ -= 8;
    }
    out_->resize (pos_);
  }
 private:
  std::vector<uint8_t>* out_;
  size_t pos_;
  uint64_t acc_ = 0;
  int nbits_ = 0;                                // bits waiting in acc_ (< 32 between calls)
};

// Append start code + NAL header + escaped RBSP to `bs`; returns the NAL length in bytes.
inline int append_nal (std::vector<uint8_t>& bs, int nal_ref_idc, int nal_type, const std::vector<uint8_t>& rbsp) {
  const size_t start = bs.size();
  bs.resize (start + 5 + rbsp.size() + rbsp.size() / 2 + 1);      // worst case: an emulation prevention byte after every second payload byte
  uint8_t* d = bs.data() + start;
  d[0] = 0; d[1] = 0; d[2] = 0; d[3] = 1;
  d[4] = (uint8_t) ((nal_ref_idc << 5) | (nal_type & 31));
  d += 5;
  int zeros = 0;
  for (const uint8_t* s = rbsp.data(), *e = s + rbsp.size(); s < e; ++s) {
    const uint8_t b = *s;
    if (zeros == 2 && b <= 3) { *d++ = 3; zeros = 0; }
    zeros = (b == 0) ? zeros + 1 : 0;
    *d++ = b;
  }
  bs.resize ((size_t) (d - bs.data()));
  return (int) (bs.size() - start);
}

}  // namespace wh
