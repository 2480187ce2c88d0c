// ist_webp.h — WebP decode (host): container + lossless in ist_webp.cpp, the lossy VP8 key frame in ist_webp_vp8.cpp
#ifndef IST_WEBP_H_
#define IST_WEBP_H_

#include <cstddef>
#include <cstdint>
#include <vector>

namespace ist {

bool is_webp(const uint8_t* file, int64_t len);
int webp_info(const uint8_t* file, int64_t len, int32_t* w, int32_t* h, int32_t* orientation);
int webp_decode_rgba8(const uint8_t* file, int64_t len, uint8_t* out, size_t pitch, int64_t out_rows);

// the payload of a "VP8 " chunk: one key frame
int vp8_info(const uint8_t* d, size_t n, int* w, int* h);
int vp8_decode_rgba8(const uint8_t* d, size_t n, uint8_t* out, size_t pitch);      // alpha = 255
// the payload of an "ALPH" chunk -> w*h alpha bytes
int webp_alpha_plane(const uint8_t* d, size_t n, int w, int h, std::vector<uint8_t>* out);

}  // namespace ist

#endif  // IST_WEBP_H_
