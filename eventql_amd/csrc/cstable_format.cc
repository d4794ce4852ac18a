#include "cstable_format.h"
#include "sha1.h"
#include <cstdio>
#include <cstring>

namespace evql {

static const uint8_t kMagic[4] = {0x23, 0x17, 0x23, 0x17};

uint32_t bitpack_width(uint32_t v) {
  uint32_t b = 0;
  while (v) {
    ++b;
    v >>= 1;
  }
  return b;
}

const ColumnLayout* TableLayout::find(const std::string& name) const {
  for (const auto& c : columns) {
    if (c.name == name) return &c;
  }
  return nullptr;
}

// ---------------------------------------------------------------------------
// parsing
// ---------------------------------------------------------------------------
namespace {
struct Cursor {
  const uint8_t* p;
  size_t len;
  size_t pos;
  bool ok;

  uint64_t fixed(int nbytes) {
    if (pos + nbytes > len) {
      ok = false;
      return 0;
    }
    uint64_t v = 0;
    for (int i = 0; i < nbytes; ++i) v |= uint64_t(p[pos + i]) << (8 * i);
    pos += nbytes;
    return v;
  }
  uint64_t varuint() {
    uint64_t v = 0;
    for (int i = 0; i < 10; ++i) {
      if (pos >= len) {
        ok = false;
        return 0;
      }
      uint8_t b = p[pos++];
      v |= uint64_t(b & 0x7f) << (7 * i);
      if (!(b & 0x80)) return v;
    }
    ok = false;
    return 0;
  }
  std::string lenenc_string() {
    uint64_t n = varuint();
    if (!ok || pos > len || n > len - pos) {  // (pos + n may wrap)
      ok = false;
      return std::string();
    }
    std::string s(reinterpret_cast<const char*>(p + pos), n);
    pos += n;
    return s;
  }
};
}  // namespace

std::string parse_cstable(const uint8_t* image, size_t len, TableLayout* out) {
  if (len < kSectorSize || memcmp(image, kMagic, 4) != 0) {
    return "not a valid cstable file";
  }
  uint16_t version = uint16_t(image[4]) | (uint16_t(image[5]) << 8);
  if (version != 2) {
    char buf[64];
    snprintf(buf, sizeof(buf), "unsupported cstable version: %u", version);
    return buf;
  }
  out->version = 2;
  out->columns.clear();

  // metablocks: the highest valid transaction id wins
  bool have_mb = false;
  for (int i = 0; i < 2; ++i) {
    const uint8_t* mb = image + kMetaBlockPosition + i * kMetaBlockSize;
    Sha1Digest h = sha1(mb, kMetaBlockSize - 20);
    if (memcmp(h.bytes, mb + kMetaBlockSize - 20, 20) != 0) continue;
    Cursor c{mb, kMetaBlockSize, 0, true};
    uint64_t txid = c.fixed(8);
    uint64_t nrows = c.fixed(8);
    uint64_t ioff = c.fixed(8);
    uint32_t isize = uint32_t(c.fixed(4));
    if (!have_mb || txid > out->transaction_id) {
      out->transaction_id = txid;
      out->num_rows = nrows;
      out->index_offset = ioff;
      out->index_size = isize;
      have_mb = true;
    }
  }
  if (!have_mb) return "can't open cstable: no valid metablocks found";

  Cursor c{image, len, size_t(kMetaBlockPosition + 2 * kMetaBlockSize + 128),
           true};
  uint64_t ncols = c.varuint();
  for (uint64_t i = 0; c.ok && i < ncols; ++i) {
    ColumnLayout col;
    col.logical_type = ColumnType(c.varuint());
    col.storage_type = ColumnEncoding(c.varuint());
    col.column_id = c.varuint();
    col.name = c.lenenc_string();
    col.rlevel_max = uint32_t(c.varuint());
    col.dlevel_max = uint32_t(c.varuint());
    out->columns.push_back(col);
  }
  if (!c.ok) return "corrupt cstable header";

  // (u64 sums of file-supplied numbers may wrap: compare without adding)
  if (out->index_offset > len || out->index_size > len - out->index_offset) {
    return "corrupt cstable: index out of bounds";
  }
  Cursor ic{image + out->index_offset, out->index_size, 0, true};
  uint64_t nentries = ic.varuint();
  for (uint64_t i = 0; ic.ok && i < nentries; ++i) {
    uint64_t kind = ic.varuint();
    uint64_t cid = ic.varuint();
    PageRef pr;
    pr.offset = ic.varuint();
    pr.size = uint32_t(ic.varuint());
    if (!ic.ok) break;
    if (pr.offset > len || pr.size > len - pr.offset) return "corrupt cstable: page out of bounds";
    for (auto& col : out->columns) {
      if (col.column_id != cid) continue;
      switch (PageKind(kind)) {
        case PageKind::DATA:
          col.data_pages.push_back(pr);
          break;
        case PageKind::RLEVEL:
          col.rlevel_pages.push_back(pr);
          break;
        case PageKind::DLEVEL:
          col.dlevel_pages.push_back(pr);
          break;
      }
    }
  }
  if (!ic.ok) return "corrupt cstable index";
  return validate_layout(image, len, *out);
}

// The kernels (and the host-side string walk) address pages by the fixed geometry
// the reference's writers produce -- 512 KiB PLAIN / LEB128 / STRING pages
// (page_writer_uint64.h:34), bit-packed pages of 1024 blocks with the u32 max_value
// in front of the first (page_writer_bitpacked.cc:43-60) -- and flat required
// streams by row number.  A file that breaks these would send them out of bounds;
// the reference's readers fail on such files with "end of column reached".
std::string validate_layout(const uint8_t* image, size_t len, const TableLayout& t) {
  (void) len;
  auto bitpacked_capacity = [&](const std::vector<PageRef>& pages, uint64_t* cap) -> bool {
    *cap = ~0ull;  // width 0: every value is 0, no pages needed
    if (pages.empty()) return true;
    if (pages[0].size < 4) return false;
    uint32_t maxv;
    memcpy(&maxv, image + pages[0].offset, 4);
    const uint32_t b = bitpack_width(maxv);
    if (b == 0) return false;  // a width-0 stream has no pages
    const uint64_t page_bytes = 16ull * b * kBitpackBlocksPerPage;
    for (size_t i = 0; i < pages.size(); ++i) {
      if (pages[i].size != page_bytes + (i == 0 ? 4 : 0)) return false;
    }
    *cap = uint64_t(pages.size()) * kBitpackBlocksPerPage * 128;
    return true;
  };
  for (const auto& c : t.columns) {
    uint64_t cap = 0;
    const bool flat_required = c.rlevel_max == 0 && c.dlevel_max == 0;
    switch (c.storage_type) {
      case ColumnEncoding::UINT32_BITPACKED:
      case ColumnEncoding::BOOLEAN_BITPACKED:
        if (!bitpacked_capacity(c.data_pages, &cap)) {
          return "corrupt cstable: bad bit-packed page geometry in column " + c.name;
        }
        break;
      default: {
        for (const auto& p : c.data_pages) {
          if (p.size != kPlainPageSize) {
            return "corrupt cstable: bad page size in column " + c.name;
          }
        }
        const uint64_t per_page =
            c.storage_type == ColumnEncoding::UINT32_PLAIN ? kPlainPageSize / 4 : kPlainPageSize / 8;
        cap = uint64_t(c.data_pages.size()) * per_page;
        if (c.storage_type == ColumnEncoding::UINT64_LEB128 ||
            c.storage_type == ColumnEncoding::STRING_PLAIN) {
          // at least one byte per value
          cap = uint64_t(c.data_pages.size()) * kPlainPageSize;
        }
      }
    }
    if (flat_required && cap < t.num_rows) return "corrupt cstable: end of column reached: " + c.name;
    uint64_t lcap = 0;
    if (!bitpacked_capacity(c.rlevel_pages, &lcap) || !bitpacked_capacity(c.dlevel_pages, &lcap)) {
      return "corrupt cstable: bad level page geometry in column " + c.name;
    }
    // (lcap is the definition level stream's here)
    if (c.rlevel_max == 0 && c.dlevel_max > 0 && lcap < t.num_rows) {
      return "corrupt cstable: end of column reached: " + c.name;
    }
  }
  return std::string();
}

// ---------------------------------------------------------------------------
// bit packing (libsimdcomp 4-lane vertical layout)
// ---------------------------------------------------------------------------
void simd_pack128(const uint32_t* in, uint32_t b, uint8_t* out) {
  uint32_t words[128];
  memset(words, 0, sizeof(uint32_t) * 4 * b);
  const uint64_t mask = b >= 32 ? 0xffffffffull : ((1ull << b) - 1);
  for (uint32_t i = 0; i < 128; ++i) {
    uint32_t lane = i & 3, k = i >> 2;
    uint32_t p = k * b, w = p >> 5, s = p & 31;
    uint64_t v = uint64_t(in[i]) & mask;
    words[4 * w + lane] |= uint32_t(v << s);
    if (s + b > 32) words[4 * (w + 1) + lane] |= uint32_t(v >> (32 - s));
  }
  memcpy(out, words, 16 * b);
}

void simd_unpack128(const uint8_t* in, uint32_t b, uint32_t* out) {
  uint32_t words[128 + 4];
  memset(words, 0, sizeof(words));
  memcpy(words, in, 16 * b);
  const uint64_t mask = b >= 32 ? 0xffffffffull : ((1ull << b) - 1);
  for (uint32_t i = 0; i < 128; ++i) {
    uint32_t lane = i & 3, k = i >> 2;
    uint32_t p = k * b, w = p >> 5, s = p & 31;
    uint64_t v = uint64_t(words[4 * w + lane]) >> s;
    if (s + b > 32) v |= uint64_t(words[4 * (w + 1) + lane]) << (32 - s);
    out[i] = uint32_t(v & mask);
  }
}

// ---------------------------------------------------------------------------
// writer
// ---------------------------------------------------------------------------
namespace {
void put_fixed(std::vector<uint8_t>* b, uint64_t v, int n) {
  for (int i = 0; i < n; ++i) b->push_back(uint8_t(v >> (8 * i)));
}
void put_varuint(std::vector<uint8_t>* b, uint64_t v) {
  do {
    uint8_t x = v & 0x7f;
    v >>= 7;
    if (v) x |= 0x80;
    b->push_back(x);
  } while (v);
}
}  // namespace

TableWriter::TableWriter(const std::vector<ColumnSpec>& columns) {
  // header (cstable.cc:173-198)
  put_fixed(&image_, 0x17231723u, 4);
  put_fixed(&image_, 2, 2);
  put_fixed(&image_, 0, 8);
  image_.resize(image_.size() + 2 * kMetaBlockSize + 128, 0);
  put_varuint(&image_, columns.size());
  for (const auto& c : columns) {
    put_varuint(&image_, uint8_t(c.logical_type));
    put_varuint(&image_, uint8_t(c.storage_type));
    put_varuint(&image_, c.column_id);
    put_varuint(&image_, c.name.size());
    image_.insert(image_.end(), c.name.begin(), c.name.end());
    put_varuint(&image_, c.rlevel_max);
    put_varuint(&image_, c.dlevel_max);
  }
  size_t padded = (image_.size() + kSectorSize - 1) / kSectorSize * kSectorSize;
  image_.resize(padded, 0);
  allocated_ = padded;

  for (const auto& c : columns) {
    Col col;
    col.spec = c;
    if (c.rlevel_max > 0) {
      col.rlevel.enabled = true;
      col.rlevel.kind = PageKind::RLEVEL;
      col.rlevel.column_id = c.column_id;
      col.rlevel.mode = 2;
      col.rlevel.max_value = c.rlevel_max;
      col.rlevel.maxbits = bitpack_width(c.rlevel_max);
    }
    if (c.dlevel_max > 0) {
      col.dlevel.enabled = true;
      col.dlevel.kind = PageKind::DLEVEL;
      col.dlevel.column_id = c.column_id;
      col.dlevel.mode = 2;
      col.dlevel.max_value = c.dlevel_max;
      col.dlevel.maxbits = bitpack_width(c.dlevel_max);
    }
    col.data.enabled = true;
    col.data.kind = PageKind::DATA;
    col.data.column_id = c.column_id;
    switch (c.storage_type) {
      case ColumnEncoding::UINT64_PLAIN:
      case ColumnEncoding::FLOAT_IEEE754:
        col.data.mode = 0;
        break;
      case ColumnEncoding::UINT32_PLAIN:
        col.data.mode = 1;
        break;
      case ColumnEncoding::UINT32_BITPACKED:
      case ColumnEncoding::BOOLEAN_BITPACKED:
        col.data.mode = 2;
        col.data.max_value = c.bitpack_max_value;
        col.data.maxbits = bitpack_width(c.bitpack_max_value);
        break;
      case ColumnEncoding::UINT64_LEB128:
      case ColumnEncoding::STRING_PLAIN:
        col.data.mode = 3;
        break;
    }
    cols_.push_back(col);
  }
}

int TableWriter::column_index(const std::string& name) const {
  for (size_t i = 0; i < cols_.size(); ++i) {
    if (cols_[i].spec.name == name) return int(i);
  }
  return -1;
}

void TableWriter::alloc_page(Stream* s, uint32_t size) {
  // PageManager::allocPage (page_manager.cc:50-74): zero-filled, appended at
  // the running offset, recorded in the index in allocation order
  s->page_off = allocated_;
  s->page_size = size;
  s->page_pos = 0;
  s->has_page = true;
  allocated_ += size;
  image_.resize(allocated_, 0);
  index_.push_back({s->kind, s->column_id, {s->page_off, size}});
}

void TableWriter::append_u64(Stream* s, uint64_t v) {
  if (!s->has_page || s->page_pos + 8 > s->page_size) {
    alloc_page(s, kPlainPageSize);
  }
  memcpy(&image_[s->page_off + s->page_pos], &v, 8);
  s->page_pos += 8;
}

void TableWriter::append_u32(Stream* s, uint32_t v) {
  if (!s->has_page || s->page_pos + 4 > s->page_size) {
    alloc_page(s, kPlainPageSize);
  }
  memcpy(&image_[s->page_off + s->page_pos], &v, 4);
  s->page_pos += 4;
}

void TableWriter::append_bitpacked(Stream* s, uint64_t v) {
  // BitPackedIntPageWriter::appendValue (page_writer_bitpacked.cc:43-67)
  if (s->maxbits == 0) return;
  if (!s->has_page) {
    alloc_page(s, 4 + 16 * s->maxbits * kBitpackBlocksPerPage);
    memcpy(&image_[s->page_off], &s->max_value, 4);
    s->page_pos = 4;
  } else if (s->page_pos >= s->page_size) {
    alloc_page(s, 16 * s->maxbits * kBitpackBlocksPerPage);
  }
  s->inbuf[s->inbuf_size++] = uint32_t(v);
  if (s->inbuf_size == 128) {
    flush_bitpacked(s);
    s->inbuf_size = 0;
    s->page_pos += 16 * s->maxbits;
  }
}

void TableWriter::flush_bitpacked(Stream* s) {
  if (s->inbuf_size == 0) return;
  for (uint32_t i = s->inbuf_size; i < 128; ++i) s->inbuf[i] = 0;
  simd_pack128(s->inbuf, s->maxbits, &image_[s->page_off + s->page_pos]);
}

void TableWriter::append_bytes(Stream* s, const uint8_t* p, size_t n) {
  // LEB128PageWriter / LenencStringPageWriter: bytes may straddle pages
  while (n > 0) {
    if (!s->has_page || s->page_pos >= s->page_size) {
      alloc_page(s, kPlainPageSize);
    }
    size_t w = s->page_size - s->page_pos;
    if (w > n) w = n;
    memcpy(&image_[s->page_off + s->page_pos], p, w);
    s->page_pos += w;
    p += w;
    n -= w;
  }
}

void TableWriter::append_leb128(Stream* s, uint64_t v) {
  uint8_t buf[10];
  size_t n = 0;
  do {
    buf[n] = v & 0x7f;
    v >>= 7;
    if (v) buf[n] |= 0x80;
    ++n;
  } while (v);
  append_bytes(s, buf, n);
}

void TableWriter::write_levels(Col* c, uint64_t rlvl, uint64_t dlvl) {
  if (c->rlevel.enabled) append_bitpacked(&c->rlevel, rlvl);
  if (c->dlevel.enabled) append_bitpacked(&c->dlevel, dlvl);
}

void TableWriter::put_null(size_t col, uint64_t rlvl, uint64_t dlvl) {
  write_levels(&cols_[col], rlvl, dlvl);
}

void TableWriter::put_uint(size_t col, uint64_t rlvl, uint64_t dlvl,
                           uint64_t v) {
  Col* c = &cols_[col];
  write_levels(c, rlvl, dlvl);
  switch (c->data.mode) {
    case 0:
      append_u64(&c->data, v);
      break;
    case 1:
      append_u32(&c->data, uint32_t(v));
      break;
    case 2:
      append_bitpacked(&c->data, v);
      break;
    case 3:
      append_leb128(&c->data, v);
      break;
  }
}

void TableWriter::put_float(size_t col, uint64_t rlvl, uint64_t dlvl,
                            double v) {
  Col* c = &cols_[col];
  write_levels(c, rlvl, dlvl);
  uint64_t bits;
  memcpy(&bits, &v, 8);
  append_u64(&c->data, bits);
}

void TableWriter::put_string(size_t col, uint64_t rlvl, uint64_t dlvl,
                             const char* s, size_t len) {
  Col* c = &cols_[col];
  write_levels(c, rlvl, dlvl);
  append_leb128(&c->data, len);
  append_bytes(&c->data, reinterpret_cast<const uint8_t*>(s), len);
}

void TableWriter::commit(uint64_t num_rows) {
  if (committed_) return;
  committed_ = true;
  for (auto& c : cols_) {
    if (c.rlevel.enabled) flush_bitpacked(&c.rlevel);
    if (c.dlevel.enabled) flush_bitpacked(&c.dlevel);
    if (c.data.mode == 2) flush_bitpacked(&c.data);
  }
  // index at allocated_bytes (cstable_file.cc:136-146, cstable.cc:229-243)
  uint64_t index_offset = allocated_;
  std::vector<uint8_t> idx;
  put_varuint(&idx, index_.size());
  for (const auto& e : index_) {
    put_varuint(&idx, uint8_t(e.kind));
    put_varuint(&idx, e.column_id);
    put_varuint(&idx, e.page.offset);
    put_varuint(&idx, e.page.size);
  }
  image_.insert(image_.end(), idx.begin(), idx.end());

  // metablock, transaction 1 -> slot 1 (cstable_file.cc:148-184)
  std::vector<uint8_t> mb;
  const uint64_t txid = 1;
  put_fixed(&mb, txid, 8);
  put_fixed(&mb, num_rows, 8);
  put_fixed(&mb, index_offset, 8);
  put_fixed(&mb, idx.size(), 4);
  Sha1Digest h = sha1(mb.data(), mb.size());
  mb.insert(mb.end(), h.bytes, h.bytes + 20);
  memcpy(&image_[kMetaBlockPosition + kMetaBlockSize * (txid % 2)], mb.data(),
         kMetaBlockSize);
}

std::string TableWriter::write_file(const std::string& path) const {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return "can't open file for writing: " + path;
  size_t n = fwrite(image_.data(), 1, image_.size(), f);
  fclose(f);
  if (n != image_.size()) return "short write: " + path;
  return std::string();
}

}  // namespace evql
