// oracle/ref_dec_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Decodes an Annex-B H.264 stream with the *reference's own* decoder (ISVCDecoder,
// codec/api/wels/codec_api.h:345-470) and writes the cropped I420 frames.  Used (a) to make the
// 720p/1080p test inputs out of the reference's res/*.264 clips and (b) to
// check that a bitstream written by our encoder decodes to exactly our reconstructed frames.
//
//   ref_dec in.264 out.yuv
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "codec_api.h"

static void write_frame (FILE* fo, unsigned char* dst[3], const SBufferInfo& bi) {
  const SSysMEMBuffer& b = bi.UsrData.sSystemBuffer;
  for (int y = 0; y < b.iHeight; ++y) std::fwrite (dst[0] + (size_t)y * b.iStride[0], 1, b.iWidth, fo);
  for (int p = 1; p < 3; ++p)
    for (int y = 0; y < b.iHeight / 2; ++y) std::fwrite (dst[p] + (size_t)y * b.iStride[1], 1, b.iWidth / 2, fo);
}

int main (int argc, char** argv) {
  if (argc < 3) { std::fprintf (stderr, "usage: ref_dec in.264 out.yuv\n"); return 2; }
  FILE* fi = std::fopen (argv[1], "rb");
  if (!fi) { std::fprintf (stderr, "cannot open %s\n", argv[1]); return 1; }
  std::fseek (fi, 0, SEEK_END); long sz = std::ftell (fi); std::fseek (fi, 0, SEEK_SET);
  std::vector<unsigned char> bs ((size_t)sz + 4);
  if (std::fread (bs.data(), 1, sz, fi) != (size_t)sz) return 1;
  std::fclose (fi);
  FILE* fo = std::fopen (argv[2], "wb");
  if (!fo) return 1;

  ISVCDecoder* dec = NULL;
  if (WelsCreateDecoder (&dec) || !dec) return 1;
  SDecodingParam dp; std::memset (&dp, 0, sizeof (dp));
  dp.uiTargetDqLayer = (unsigned char) - 1;
  dp.eEcActiveIdc = ERROR_CON_DISABLE;
  dp.sVideoProperty.eVideoBsType = VIDEO_BITSTREAM_DEFAULT;
  if (dec->Initialize (&dp)) return 1;

  // split at access-unit granularity is not needed: feed one NAL at a time (start-code scan)
  int frames = 0;
  long pos = 0;
  auto is_sc = [&] (long p) { return p + 3 < sz && bs[p] == 0 && bs[p + 1] == 0 && ((bs[p + 2] == 1) || (bs[p + 2] == 0 && bs[p + 3] == 1)); };
  while (pos < sz) {
    long q = pos + 3;
    while (q < sz && !is_sc (q)) ++q;
    if (q >= sz) q = sz;
    unsigned char* dst[3] = {0, 0, 0};
    SBufferInfo bi; std::memset (&bi, 0, sizeof (bi));
    dec->DecodeFrame2 (bs.data() + pos, (int) (q - pos), dst, &bi);
    if (bi.iBufferStatus == 1) { write_frame (fo, dst, bi); ++frames; }
    pos = q;
  }
  for (;;) {   // drain
    unsigned char* dst[3] = {0, 0, 0};
    SBufferInfo bi; std::memset (&bi, 0, sizeof (bi));
    dec->DecodeFrame2 (NULL, 0, dst, &bi);
    if (bi.iBufferStatus != 1) break;
    write_frame (fo, dst, bi); ++frames;
  }
  int32_t left = 0;
  dec->GetOption (DECODER_OPTION_NUM_OF_FRAMES_REMAINING_IN_BUFFER, &left);
  for (int i = 0; i < left; ++i) {
    unsigned char* dst[3] = {0, 0, 0};
    SBufferInfo bi; std::memset (&bi, 0, sizeof (bi));
    dec->FlushFrame (dst, &bi);
    if (bi.iBufferStatus == 1) { write_frame (fo, dst, bi); ++frames; }
  }
  std::fclose (fo);
  dec->Uninitialize();
  WelsDestroyDecoder (dec);
  std::printf ("frames=%d\n", frames);
  return 0;
}
