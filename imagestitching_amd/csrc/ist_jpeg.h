// ist_jpeg.h — what the host-side JPEG entropy decoder hands to the GPU stages
#ifndef IST_JPEG_H_
#define IST_JPEG_H_

#include <cstdint>
#include <vector>

namespace ist {

struct JpegComp {
  int id = 0, h = 1, v = 1, tq = 0;
  int blocks_x = 0, blocks_y = 0;      // padded to whole MCUs
  uint16_t q[64];                      // quantisation table, natural (row-major) order
  std::vector<int16_t> coef;           // blocks_y * blocks_x blocks of 64 coefficients, natural order, NOT dequantised
};

struct JpegImage {
  int width = 0, height = 0, ncomp = 0;
  int hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
  int orientation = 0;                 // EXIF 1..8, 0 = absent
  int scans = 0;
  JpegComp comp[3];
};

// container parsing + Huffman decoding (host).  header_only stops after the frame header (size, sampling, orientation
// if the EXIF segment precedes it).  Returns IST_OK or an error code with the thread-local message set.
int jpeg_parse_and_entropy_decode(const uint8_t* file, int64_t len, JpegImage* out, bool header_only);

// GPU stages (ist_jpeg_kernels.hip): coefficient planes (device) -> RGBA8 (device).  d_coef[c] / d_q[c] per component,
// planes = scratch for the reconstructed sample planes.  Asynchronous on `stream`.
struct JpegDeviceJob {
  int width, height, ncomp, hmax, vmax;
  int h[3], v[3], blocks_x[3], blocks_y[3];
  const int16_t* d_coef[3];
  const uint16_t* d_q[3];
  uint8_t* d_plane[3];                 // blocks_x*8 bytes per row
  uint8_t* out; size_t out_pitch;
};
int jpeg_launch_reconstruct(const JpegDeviceJob& job, void* stream);

}  // namespace ist

#endif  // IST_JPEG_H_
