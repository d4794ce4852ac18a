// cstable v0.2.0 container: layout parser + writer (host side).
//
// This is the host half of the decode path: it locates the column pages that
// the HIP kernels read directly out of the file image in HBM.  It follows the
// grammar in the reference's io/cstable/cstable.h:67-111 and the behaviour of
//   readHeader      io/cstable/cstable.cc:35-85,200-227
//   readMetaBlock   io/cstable/cstable.cc:152-171
//   readIndex       io/cstable/cstable.cc:245-255
//   PageManager     io/cstable/page_manager.cc:44-170   (allocation order)
//   *PageWriter     io/cstable/columns/page_writer_*.cc (page fill rules)
//   CSTableWriter   io/cstable/cstable_writer.cc:267-293 (commit sequence)
// It is a fresh implementation; no reference code is included or linked.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace evql {

enum class ColumnType : uint8_t {
  SUBRECORD = 0,
  BOOLEAN = 1,
  UNSIGNED_INT = 2,
  SIGNED_INT = 3,
  STRING = 4,
  FLOAT = 5,
  DATETIME = 6
};

enum class ColumnEncoding : uint8_t {
  BOOLEAN_BITPACKED = 1,
  UINT32_BITPACKED = 10,
  UINT32_PLAIN = 11,
  UINT64_PLAIN = 12,
  UINT64_LEB128 = 13,
  FLOAT_IEEE754 = 14,
  STRING_PLAIN = 100
};

enum class PageKind : uint8_t { DATA = 1, RLEVEL = 2, DLEVEL = 3 };

struct PageRef {
  uint64_t offset;
  uint32_t size;
};

static const uint32_t kPlainPageSize = 512 * 1024;  // page_writer_uint64.h:34
static const uint32_t kBitpackBlocksPerPage = 1024;  // page_writer_bitpacked.h:36
static const uint32_t kSectorSize = 512;
static const uint32_t kMetaBlockPosition = 14;
static const uint32_t kMetaBlockSize = 48;

struct ColumnLayout {
  std::string name;
  ColumnType logical_type;
  ColumnEncoding storage_type;
  uint64_t column_id;
  uint32_t rlevel_max;
  uint32_t dlevel_max;
  std::vector<PageRef> data_pages;
  std::vector<PageRef> rlevel_pages;
  std::vector<PageRef> dlevel_pages;
};

struct TableLayout {
  uint32_t version;  // 2
  uint64_t num_rows;
  uint64_t transaction_id;
  uint64_t index_offset;
  uint32_t index_size;
  std::vector<ColumnLayout> columns;

  const ColumnLayout* find(const std::string& name) const;
};

// number of bits needed for max_value (libsimdcomp bits(), simdcomputil.c:9-20)
uint32_t bitpack_width(uint32_t max_value);

// Parses a v0.2.0 image.  Returns "" on success, else an error message.
std::string parse_cstable(const uint8_t* image, size_t len, TableLayout* out);
// page geometry / capacity checks of a parsed layout ("" = fine); parse_cstable runs it
std::string validate_layout(const uint8_t* image, size_t len, const TableLayout& t);

// Re-encodes a v0.1.0 image as v0.2.0 (cstable_v1.cc).  Returns "" on success.
std::string transcode_v1_to_v2(const uint8_t* image, size_t len, std::vector<uint8_t>* out);

// ---------------------------------------------------------------------------
// writer
// ---------------------------------------------------------------------------
struct ColumnSpec {
  std::string name;
  ColumnType logical_type;
  ColumnEncoding storage_type;
  uint64_t column_id;
  uint32_t rlevel_max;
  uint32_t dlevel_max;
  // bit width source for *_BITPACKED data streams.  The reference writer always
  // uses 0xffffffff for UINT32_BITPACKED and 1 for BOOLEAN_BITPACKED
  // (column_writer_uint.cc:57-63); narrower values are legal for the reader.
  uint32_t bitpack_max_value;
};

class TableWriter {
 public:
  explicit TableWriter(const std::vector<ColumnSpec>& columns);

  size_t num_columns() const { return cols_.size(); }
  int column_index(const std::string& name) const;

  void put_null(size_t col, uint64_t rlvl, uint64_t dlvl);
  void put_uint(size_t col, uint64_t rlvl, uint64_t dlvl, uint64_t v);
  void put_float(size_t col, uint64_t rlvl, uint64_t dlvl, double v);
  void put_string(size_t col, uint64_t rlvl, uint64_t dlvl, const char* s,
                  size_t len);

  // flush all streams, append index, write metablock (txid 1, slot txid % 2)
  void commit(uint64_t num_rows);

  const std::vector<uint8_t>& image() const { return image_; }
  std::string write_file(const std::string& path) const;  // "" on success

 private:
  struct Stream {
    PageKind kind;
    uint64_t column_id;
    // 0 = u64 plain, 1 = u32 plain, 2 = bitpacked, 3 = leb128/bytes
    int mode;
    bool has_page = false;
    uint64_t page_off = 0;
    uint32_t page_size = 0;
    uint64_t page_pos = 0;
    // bitpacked state
    uint32_t max_value = 0;
    uint32_t maxbits = 0;
    uint32_t inbuf[128];
    uint32_t inbuf_size = 0;
    bool enabled = false;
  };
  struct Col {
    ColumnSpec spec;
    Stream rlevel, dlevel, data;
  };
  struct IndexEntry {
    PageKind kind;
    uint64_t column_id;
    PageRef page;
  };

  void alloc_page(Stream* s, uint32_t size);
  void append_u64(Stream* s, uint64_t v);
  void append_u32(Stream* s, uint32_t v);
  void append_bitpacked(Stream* s, uint64_t v);
  void flush_bitpacked(Stream* s);
  void append_bytes(Stream* s, const uint8_t* p, size_t n);
  void append_leb128(Stream* s, uint64_t v);
  void write_levels(Col* c, uint64_t rlvl, uint64_t dlvl);

  std::vector<Col> cols_;
  std::vector<IndexEntry> index_;
  std::vector<uint8_t> image_;
  uint64_t allocated_;
  bool committed_ = false;
};

// packs 128 values into 16*b bytes using libsimdcomp's 4-lane vertical layout
// (simdbitpacking.c)
void simd_pack128(const uint32_t* in, uint32_t b, uint8_t* out);
void simd_unpack128(const uint8_t* in, uint32_t b, uint32_t* out);

}  // namespace evql
