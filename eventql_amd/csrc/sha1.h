// SHA-1 (FIPS 180-4).  Needed on the host side of the path for two things the
// reference does with it:
//   * cstable v0.2.0 metablock checksum  (reference: io/cstable/cstable.cc:138-171)
//   * GroupBy group identity / partial-aggregate wire key
//     (reference: sql/statements/select/groupby.cc:129-135, util/SHA1.h)
#pragma once
#include <cstddef>
#include <cstdint>

namespace evql {

struct Sha1Digest {
  uint8_t bytes[20];
};

Sha1Digest sha1(const void* data, size_t len);

}  // namespace evql
