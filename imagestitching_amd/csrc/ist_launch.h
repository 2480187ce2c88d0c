// ist_launch.h — kernel argument block + launch entry shared by ist_kernels.hip and ist_runtime.cpp
#ifndef IST_LAUNCH_H_
#define IST_LAUNCH_H_

#include "ist_internal.h"

namespace ist {

// Source pointers travel BY VALUE in the kernarg segment (2 KiB for 128 images): a stitch can be re-launched on
// new buffers with no table upload and no host synchronisation.  The reference UI caps a stitch at 9 images
// (pages/index/index.js:311); BASELINE config 5 uses 64.
constexpr int kMaxImages = 128;

struct LaunchArgs {
  uint8_t* dst;
  size_t dst_pitch;
  const DevOp* ops;
  const DevCell* cells;
  const DevBand* bands;
  const int32_t* stacks;
  const DevTile* tiles;     // per-tile table, or NULL (then the band / cell prefixes are searched)
  int32_t n_cells;
  int32_t filter;
  int32_t lds_words;        // dynamic LDS the SAMPLE_LDS cells need (32-bit words); 0 when the job has none
  int32_t n_bands;
  int32_t lds_half;         // SAMPLE_LDS footprint buffer (32-bit words)
  int32_t pad_;
  const uint8_t* src[kMaxImages];
  size_t pitch[kMaxImages];
};

// kind: 0 = the job has only FILL / COPY cells, 1 = + axis-aligned resampling (SAMPLE, SAMPLE_LDS, SAMPLE_STREAM), 3 = + the
// streamed box filter (AREA_STREAM), 2 = quarter turns / paint stacks (no box filter), 4 = everything: the instantiation
// with just those paths is launched
int launch_stitch(const LaunchArgs& args, int64_t n_tiles, int kind, void* stream);

}  // namespace ist

#endif  // IST_LAUNCH_H_
