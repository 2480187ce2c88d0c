// placeholder replaced below
#include "ist_internal.h"
#include "ist_webp.h"
namespace ist {
int vp8_info(const uint8_t* d, size_t n, int* w, int* h) {
  if (n < 10 || (d[0] & 1) != 0 || d[3] != 0x9D || d[4] != 0x01 || d[5] != 0x2A) return fail(IST_E_DECODE, "WebP: not a VP8 key frame");
  *w = (d[6] | (d[7] << 8)) & 0x3FFF; *h = (d[8] | (d[9] << 8)) & 0x3FFF;
  if (*w < 1 || *h < 1) return fail(IST_E_DECODE, "WebP: bad VP8 frame size");
  return IST_OK;
}
int vp8_decode_rgba8(const uint8_t*, size_t, uint8_t*, size_t) { return fail(IST_E_UNSUPPORTED, "lossy WebP (VP8) is not built yet"); }
}
