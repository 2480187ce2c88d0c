// ist_crc.h — CRC-32 (PNG / zlib, reflected 0xEDB88320) helpers shared by the two PNG encoders
#ifndef IST_CRC_H_
#define IST_CRC_H_

#include <cstdint>

namespace ist {

constexpr uint32_t kCrcPoly = 0xEDB88320u;

struct CrcTables { uint32_t t[4][256]; };

// byte-at-a-time table and the three slices on top of it
inline void make_crc_tables(CrcTables* T) {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ kCrcPoly : c >> 1;
    T->t[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int s = 1; s < 4; ++s) T->t[s][i] = (T->t[s - 1][i] >> 8) ^ T->t[0][T->t[s - 1][i] & 0xFF];
}

inline uint32_t crc_byte(const CrcTables& T, uint32_t reg, uint8_t b) { return T.t[0][(reg ^ b) & 0xFF] ^ (reg >> 8); }

// product of two polynomials over GF(2) modulo the CRC polynomial, reflected bit order (x^0 = 0x80000000)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t gf_mul(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int i = 0; i < 32; ++i) {
    if (a & (0x80000000u >> i)) p ^= b;
    b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
  }
  return p;
}

}  // namespace ist

#endif  // IST_CRC_H_
