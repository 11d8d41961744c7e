// bitwriter.h -- RBSP bit writer, Exp-Golomb codes and Annex-B NAL encapsulation (host side).
//
// Syntax elements follow ITU-T H.264 7.2 / 9.1; behaviourally equivalent to the reference's
// codec/common/inc/golomb_common.h:102-163 and codec/encoder/core/src/nal_encap.cpp:118-185.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <vector>

namespace wh {

class BitWriter {
 public:
  explicit BitWriter (std::vector<uint8_t>* out) : out_ (out) {}
  void put (int n, uint32_t v) {                 // n in [0,32], MSB first
    while (n > 0) {
      const int take = n < free_ ? n : free_;    // <= 8
      const uint32_t chunk = (v >> (n - take)) & ((1u << take) - 1u);
      cur_ = (cur_ << take) | chunk;
      free_ -= take;
      n -= take;
      if (free_ == 0) { out_->push_back ((uint8_t)cur_); cur_ = 0; free_ = 8; }
    }
  }
  void bit (int b) { put (1, b ? 1u : 0u); }
  void ue (uint32_t v) {
    const uint32_t x = v + 1;
    int len = 0;
    while ((x >> len) > 1) ++len;                // floor(log2(x))
    put (len, 0);
    put (len + 1, x);
  }
  void se (int32_t v) { ue (v > 0 ? (uint32_t) (2 * v - 1) : (uint32_t) (-2 * v)); }
  void te (int max_minus1, uint32_t v) { if (max_minus1 == 1) bit (!v); else ue (v); }
  void trailing () { bit (1); while (free_ != 8) bit (0); }
  size_t bits () const { return out_->size() * 8 + (8 - free_); }
 private:
  std::vector<uint8_t>* out_;
  uint32_t cur_ = 0;
  int free_ = 8;
};

// Append start code + NAL header + escaped RBSP to `bs`; returns the NAL length in bytes.
inline int append_nal (std::vector<uint8_t>& bs, int nal_ref_idc, int nal_type, const std::vector<uint8_t>& rbsp) {
  const size_t start = bs.size();
  bs.push_back (0); bs.push_back (0); bs.push_back (0); bs.push_back (1);
  bs.push_back ((uint8_t) ((nal_ref_idc << 5) | (nal_type & 31)));
  int zeros = 0;
  for (size_t i = 0; i < rbsp.size(); ++i) {
    const uint8_t b = rbsp[i];
    if (zeros == 2 && b <= 3) { bs.push_back (3); zeros = 0; }
    zeros = (b == 0) ? zeros + 1 : 0;
    bs.push_back (b);
  }
  return (int) (bs.size() - start);
}

}  // namespace wh
