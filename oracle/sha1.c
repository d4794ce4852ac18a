/*
 * sha1.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * SHA-1 as used by the reference for metablock checksums and GROUP BY group
 * identity (reference: src/eventql/util/SHA1.cc, SHA1::compute).  Standard
 * FIPS 180-4 algorithm; pinned against the reference's own SHA1 through
 * oracle/_ref (tests/test_oracle_vs_ref.py) and RFC 3174 vectors.
 */
#include "oracle.h"
#include <string.h>

#define ROL(v, s) (((v) << (s)) | ((v) >> (32 - (s))))

static void block(uint32_t h[5], const uint8_t* p) {
  uint32_t w[80], a, b, c, d, e, f, k, t;
  int i;
  for (i = 0; i < 16; ++i) {
    w[i] = ((uint32_t) p[4 * i] << 24) | ((uint32_t) p[4 * i + 1] << 16) |
           ((uint32_t) p[4 * i + 2] << 8) | p[4 * i + 3];
  }
  for (; i < 80; ++i) {
    t = w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16];
    w[i] = ROL(t, 1);
  }
  a = h[0]; b = h[1]; c = h[2]; d = h[3]; e = h[4];
  for (i = 0; i < 80; ++i) {
    if (i < 20) { f = (b & c) | (~b & d); k = 0x5a827999u; }
    else if (i < 40) { f = b ^ c ^ d; k = 0x6ed9eba1u; }
    else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8f1bbcdcu; }
    else { f = b ^ c ^ d; k = 0xca62c1d6u; }
    t = ROL(a, 5) + f + e + k + w[i];
    e = d; d = c; c = ROL(b, 30); b = a; a = t;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e;
}

void orc_sha1(const void* data, size_t len, uint8_t out[20]) {
  uint32_t h[5] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u, 0xc3d2e1f0u};
  const uint8_t* p = (const uint8_t*) data;
  size_t n = len, tl;
  uint8_t tail[128];
  uint64_t bits = (uint64_t) len * 8;
  int i;
  for (; n >= 64; p += 64, n -= 64) block(h, p);
  memset(tail, 0, sizeof(tail));
  memcpy(tail, p, n);
  tail[n] = 0x80;
  tl = (n + 9 <= 64) ? 64 : 128;
  for (i = 0; i < 8; ++i) tail[tl - 1 - i] = (uint8_t) (bits >> (8 * i));
  block(h, tail);
  if (tl == 128) block(h, tail + 64);
  for (i = 0; i < 5; ++i) {
    out[4 * i] = (uint8_t) (h[i] >> 24);
    out[4 * i + 1] = (uint8_t) (h[i] >> 16);
    out[4 * i + 2] = (uint8_t) (h[i] >> 8);
    out[4 * i + 3] = (uint8_t) h[i];
  }
}
