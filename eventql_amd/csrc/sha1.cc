#include "sha1.h"
#include <cstring>

namespace evql {

namespace {
inline uint32_t rotl(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }

void sha1_block(uint32_t h[5], const uint8_t* p) {
  uint32_t w[80];
  for (int t = 0; t < 16; ++t) {
    w[t] = (uint32_t(p[4 * t]) << 24) | (uint32_t(p[4 * t + 1]) << 16) |
           (uint32_t(p[4 * t + 2]) << 8) | uint32_t(p[4 * t + 3]);
  }
  for (int t = 16; t < 80; ++t) {
    w[t] = rotl(w[t - 3] ^ w[t - 8] ^ w[t - 14] ^ w[t - 16], 1);
  }
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
  for (int t = 0; t < 80; ++t) {
    uint32_t f, k;
    if (t < 20) {
      f = (b & c) | (~b & d);
      k = 0x5a827999u;
    } else if (t < 40) {
      f = b ^ c ^ d;
      k = 0x6ed9eba1u;
    } else if (t < 60) {
      f = (b & c) | (b & d) | (c & d);
      k = 0x8f1bbcdcu;
    } else {
      f = b ^ c ^ d;
      k = 0xca62c1d6u;
    }
    uint32_t tmp = rotl(a, 5) + f + e + k + w[t];
    e = d;
    d = c;
    c = rotl(b, 30);
    b = a;
    a = tmp;
  }
  h[0] += a;
  h[1] += b;
  h[2] += c;
  h[3] += d;
  h[4] += e;
}
}  // namespace

Sha1Digest sha1(const void* data, size_t len) {
  uint32_t h[5] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u,
                   0xc3d2e1f0u};
  const uint8_t* p = static_cast<const uint8_t*>(data);
  size_t n = len;
  while (n >= 64) {
    sha1_block(h, p);
    p += 64;
    n -= 64;
  }
  uint8_t tail[128];
  memset(tail, 0, sizeof(tail));
  memcpy(tail, p, n);
  tail[n] = 0x80;
  size_t tl = (n + 9 <= 64) ? 64 : 128;
  uint64_t bits = uint64_t(len) * 8;
  for (int i = 0; i < 8; ++i) {
    tail[tl - 1 - i] = uint8_t(bits >> (8 * i));
  }
  sha1_block(h, tail);
  if (tl == 128) sha1_block(h, tail + 64);
  Sha1Digest out;
  for (int i = 0; i < 5; ++i) {
    out.bytes[4 * i] = uint8_t(h[i] >> 24);
    out.bytes[4 * i + 1] = uint8_t(h[i] >> 16);
    out.bytes[4 * i + 2] = uint8_t(h[i] >> 8);
    out.bytes[4 * i + 3] = uint8_t(h[i]);
  }
  return out;
}

}  // namespace evql
