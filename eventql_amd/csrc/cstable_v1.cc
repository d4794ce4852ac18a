// cstable_v1.cc -- cstable v0.1.0 files (the reference's only checked-in binary
// fixture, test/sql_testdata/testtbl.cst, is v0.1.0).  They are re-encoded on
// the host into the v0.2.0 page layout the kernels read, keeping every
// (repetition level, definition level, value) triple and every column's
// storage type.
//
// Reference: header  io/cstable/cstable.cc:89-132
//            body    io/cstable/columns/v1/ColumnReader.h:36-52
//                    [u64 nvals][u64 rlvl_size][u64 dlvl_size][u64 data_size]
//                    [rlvl bit-packed][dlvl bit-packed][data]
//            levels  util/util/BitPackDecoder.{h,cc} (simdcomp blocks, width from
//                    the header's max level, no max_value prefix)
//            data    columns/v1/{UInt32,UInt64,LEB128,BitPackedInt,Boolean,Double,
//                    String}ColumnReader.cc
#include <cstring>
#include "cstable_format.h"

namespace evql {

namespace {

uint64_t rd(const uint8_t* p, int n) {
  uint64_t v = 0;
  for (int i = 0; i < n; ++i) v |= uint64_t(p[i]) << (8 * i);
  return v;
}

// sequential reader of a raw bit-packed region (blocks of 128 values)
struct BitReader {
  const uint8_t* p;
  size_t len, pos = 0;
  uint32_t bits;
  uint32_t buf[128];
  int bufpos = 128;
  uint32_t next() {
    if (bits == 0) return 0;
    if (bufpos == 128) {
      uint8_t block[512];
      memset(block, 0, sizeof(block));
      size_t n = 16 * size_t(bits);
      if (pos < len) memcpy(block, p + pos, pos + n <= len ? n : len - pos);
      pos += n;
      simd_unpack128(block, bits, buf);
      bufpos = 0;
    }
    return buf[bufpos++];
  }
};

}  // namespace

std::string transcode_v1_to_v2(const uint8_t* image, size_t len, std::vector<uint8_t>* out) {
  if (len < 26 || image[0] != 0x23 || image[1] != 0x17 || image[2] != 0x23 || image[3] != 0x17) {
    return "not a valid cstable file";
  }
  if (rd(image + 4, 2) != 1) return "not a cstable v0.1.0 file";
  size_t pos = 6 + 8;
  const uint64_t num_rows = rd(image + pos, 8);
  pos += 8;
  const uint32_t ncols = uint32_t(rd(image + pos, 4));
  pos += 4;
  struct V1Col {
    ColumnSpec spec;
    uint64_t body_off, body_size;
  };
  std::vector<V1Col> cols;
  for (uint32_t i = 0; i < ncols; ++i) {
    if (pos + 8 > len) return "corrupt cstable v0.1.0 header";
    V1Col c;
    const uint32_t enc = uint32_t(rd(image + pos, 4));
    const uint32_t nl = uint32_t(rd(image + pos + 4, 4));
    pos += 8;
    if (pos + nl + 24 > len) return "corrupt cstable v0.1.0 header";
    c.spec.name.assign(reinterpret_cast<const char*>(image + pos), nl);
    pos += nl;
    c.spec.storage_type = ColumnEncoding(enc);
    switch (c.spec.storage_type) {
      case ColumnEncoding::BOOLEAN_BITPACKED: c.spec.logical_type = ColumnType::BOOLEAN; break;
      case ColumnEncoding::FLOAT_IEEE754: c.spec.logical_type = ColumnType::FLOAT; break;
      case ColumnEncoding::STRING_PLAIN: c.spec.logical_type = ColumnType::STRING; break;
      case ColumnEncoding::UINT32_BITPACKED:
      case ColumnEncoding::UINT32_PLAIN:
      case ColumnEncoding::UINT64_PLAIN:
      case ColumnEncoding::UINT64_LEB128: c.spec.logical_type = ColumnType::UNSIGNED_INT; break;
      default: return "unsupported column encoding in a v0.1.0 file";
    }
    c.spec.rlevel_max = uint32_t(rd(image + pos, 4));
    c.spec.dlevel_max = uint32_t(rd(image + pos + 4, 4));
    c.body_off = rd(image + pos + 8, 8);
    c.body_size = rd(image + pos + 16, 8);
    pos += 24;
    if (c.body_off + c.body_size > len || c.body_size < 32) return "corrupt cstable v0.1.0 body";
    c.spec.column_id = i + 1;
    c.spec.bitpack_max_value =
        c.spec.storage_type == ColumnEncoding::BOOLEAN_BITPACKED ? 1u : 0xffffffffu;
    cols.push_back(c);
  }
  std::vector<ColumnSpec> specs;
  for (auto& c : cols) {
    if (c.spec.storage_type == ColumnEncoding::UINT32_BITPACKED) {
      // keep the file's own width: u32 max precedes the blocks
      const uint8_t* body = image + c.body_off;
      const uint64_t rs = rd(body + 8, 8), ds = rd(body + 16, 8);
      uint32_t maxv = uint32_t(rd(body + 32 + rs + ds, 4));
      c.spec.bitpack_max_value = maxv ? maxv : 1;
    }
    specs.push_back(c.spec);
  }
  TableWriter w(specs);
  for (size_t ci = 0; ci < cols.size(); ++ci) {
    const V1Col& c = cols[ci];
    const uint8_t* body = image + c.body_off;
    const uint64_t nvals = rd(body, 8), rs = rd(body + 8, 8), ds = rd(body + 16, 8);
    const uint64_t dsz = rd(body + 24, 8);
    if (32 + rs + ds + dsz > c.body_size) return "corrupt cstable v0.1.0 body";
    BitReader rl{body + 32, size_t(rs), 0, bitpack_width(c.spec.rlevel_max)};
    BitReader dl{body + 32 + rs, size_t(ds), 0, bitpack_width(c.spec.dlevel_max)};
    const uint8_t* data = body + 32 + rs + ds;
    size_t dpos = 0;
    BitReader vals{nullptr, 0, 0, 0};
    if (c.spec.storage_type == ColumnEncoding::UINT32_BITPACKED) {
      vals = BitReader{data + 4, dsz >= 4 ? size_t(dsz - 4) : 0, 0,
                       bitpack_width(uint32_t(rd(data, 4)))};
    } else if (c.spec.storage_type == ColumnEncoding::BOOLEAN_BITPACKED) {
      vals = BitReader{data, size_t(dsz), 0, 1};
    }
    for (uint64_t i = 0; i < nvals; ++i) {
      const uint64_t r = rl.next(), d = dl.next();
      if (d != c.spec.dlevel_max) {
        w.put_null(ci, r, d);
        continue;
      }
      switch (c.spec.storage_type) {
        case ColumnEncoding::UINT32_BITPACKED:
        case ColumnEncoding::BOOLEAN_BITPACKED:
          w.put_uint(ci, r, d, vals.next());
          break;
        case ColumnEncoding::UINT32_PLAIN:
          if (dpos + 4 > dsz) return "corrupt cstable v0.1.0 data";
          w.put_uint(ci, r, d, rd(data + dpos, 4));
          dpos += 4;
          break;
        case ColumnEncoding::UINT64_PLAIN:
          if (dpos + 8 > dsz) return "corrupt cstable v0.1.0 data";
          w.put_uint(ci, r, d, rd(data + dpos, 8));
          dpos += 8;
          break;
        case ColumnEncoding::UINT64_LEB128: {
          uint64_t v = 0;
          for (int k = 0; k < 10 && dpos < dsz; ++k) {
            uint8_t b = data[dpos++];
            v |= uint64_t(b & 0x7f) << (7 * k);
            if (!(b & 0x80)) break;
          }
          w.put_uint(ci, r, d, v);
          break;
        }
        case ColumnEncoding::FLOAT_IEEE754: {
          if (dpos + 8 > dsz) return "corrupt cstable v0.1.0 data";
          uint64_t bits = rd(data + dpos, 8);
          double f;
          memcpy(&f, &bits, 8);
          w.put_float(ci, r, d, f);
          dpos += 8;
          break;
        }
        case ColumnEncoding::STRING_PLAIN: {
          if (dpos + 4 > dsz) return "corrupt cstable v0.1.0 data";
          uint32_t sl = uint32_t(rd(data + dpos, 4));
          dpos += 4;
          if (dpos + sl > dsz) return "corrupt cstable v0.1.0 data";
          w.put_string(ci, r, d, reinterpret_cast<const char*>(data + dpos), sl);
          dpos += sl;
          break;
        }
      }
    }
  }
  w.commit(num_rows);
  *out = w.image();
  return std::string();
}

}  // namespace evql
