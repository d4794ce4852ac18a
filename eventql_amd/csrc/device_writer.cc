// device_writer.cc -- cstable v0.2.0 files written on the device.
//
// The reference's write side (io/cstable/cstable_writer.cc, page_writer_*.cc,
// cstable_file.cc:136-184) appends value by value on one CPU thread.  Here the
// columns already sit in HBM as SoA arrays (compaction output, query results, a
// generator) and every stream is encoded by data-parallel passes straight into the
// table image in HBM; only the header, the page index and the metablock -- a few
// KiB -- are produced on the host.  The result is an ordinary `evql_table`, and
// `evql_table_download_image` yields the file bytes.
//
// Page placement: PageManager::allocPage (page_manager.cc:50-74) hands out pages
// at the running file offset in the order streams first need them, so the byte
// layout of a reference-written file depends on how rows of different columns
// interleave.  This writer lays a column's definition-level pages down first,
// then its data pages, column after column -- what the host writer
// (cstable_format.cc TableWriter, one whole column per `put`) produces whenever
// each stream of an optional column fits one page, and exactly for required
// columns of any size.  Readers follow the page index, so every order is valid.
//
// Repeated / nested columns arrive shredded, one (r, d, value) triple per slot: their
// level streams are bit-packed from the level arrays and the values of the slots with
// d == dlevel_max compacted like an optional column's.
//
// String columns (LenencStringPageWriter, page_writer_lenencstring.cc:37-69) arrive
// as one word per row, (length << 40) | offset into a byte heap in HBM -- the form in
// which the scan side names a string of a resident table (MaterializedColumn::d_strpos)
// -- and are encoded like LEB128 streams: sizes per 2048-value chunk, scan, bytes at
// their stream positions.
#include <cstring>
#include <memory>
#include "runtime.h"
#include "sha1.h"

namespace evql {

#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                        \
      return Status::error(EVQL_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    }                                                                              \
  } while (0)

namespace {
struct ColumnWork {
  uint64_t nslots = 0;              // level-stream length (= rows for flat columns)
  DevBuf<uint8_t> nulls_owned;      // nested: d != dlevel_max per slot
  const uint8_t* nulls = nullptr;
  DevBuf<uint64_t> rwords, dwords;  // nested: the level streams as words
  DevBuf<uint64_t> dense_owned;
  const uint64_t* dense = nullptr;  // defined values in row order
  uint64_t ndef = 0;
  DevBuf<uint64_t> chunk_offsets;   // LEB128: scanned bytes per 2048-value chunk
  uint64_t leb_bytes = 0;
  uint32_t bits = 0;                // bit-packed data stream width
};

void put_varuint(std::vector<uint8_t>* b, uint64_t v) {
  do {
    uint8_t x = v & 0x7f;
    v >>= 7;
    if (v) x |= 0x80;
    b->push_back(x);
  } while (v);
}
}  // namespace

Status table_from_device_columns(evql_ctx* ctx, const std::vector<ColumnSpec>& specs,
                                 const std::vector<DeviceColumnIn>& in, uint64_t n,
                                 evql_table** out) {
  hipStream_t s = ctx->stream;
  std::vector<ColumnWork> work(specs.size());

  // ---- pass 1: value counts and encoded sizes ------------------------------------------
  for (size_t i = 0; i < specs.size(); ++i) {
    const ColumnSpec& c = specs[i];
    ColumnWork& w = work[i];
    const bool nested = c.rlevel_max > 0 || c.dlevel_max > 1;
    w.nslots = nested ? in[i].num_slots : n;
    if (nested) {
      if (in[i].nulls || (w.nslots && (!in[i].dlevels || (c.rlevel_max > 0 && !in[i].rlevels)))) {
        return Status::error(EVQL_EARG, "device writer: a repeated / nested column takes level "
                                        "arrays per slot: " + c.name);
      }
      if (w.nslots < n) return Status::error(EVQL_EARG, "device writer: fewer slots than rows: " + c.name);
    }
    const bool is_string = c.storage_type == ColumnEncoding::STRING_PLAIN;
    if (is_string != (c.logical_type == ColumnType::STRING)) {
      return Status::error(EVQL_EARG, "device writer: STRING columns use STRING_PLAIN: " + c.name);
    }
    if (is_string && !in[i].bytes && n) {
      return Status::error(EVQL_EARG, "device writer: string column without a byte heap: " + c.name);
    }
    if (!nested && (c.dlevel_max > 0) != (in[i].nulls != nullptr)) {
      return Status::error(EVQL_EARG, "device writer: NULL flags are given exactly for optional "
                                      "columns (dlevel_max 1): " + c.name);
    }
    if (!in[i].values && w.nslots) return Status::error(EVQL_EARG, "device writer: no values: " + c.name);
    w.dense = in[i].values;
    w.ndef = w.nslots;
    w.nulls = in[i].nulls;
    if (nested && w.nslots) {
      // slots without a value (d != dlevel_max) as NULL flags; level streams as words
      HIP_TRY(w.nulls_owned.alloc(w.nslots));
      HIP_TRY(w.dwords.alloc(w.nslots * 8));
      HIP_TRY(launch_wr_levels(in[i].dlevels, w.nslots, c.dlevel_max, w.nulls_owned, w.dwords, s));
      if (c.rlevel_max > 0) {
        HIP_TRY(w.rwords.alloc(w.nslots * 8));
        HIP_TRY(launch_wr_levels(in[i].rlevels, w.nslots, 0, nullptr, w.rwords, s));
      }
      w.nulls = w.nulls_owned;
    }
    if (c.dlevel_max > 0 && w.nslots) {
      const uint64_t nt = (w.nslots + kDecodeTile - 1) / kDecodeTile;
      DevBuf<uint64_t> d_tiles;
      HIP_TRY(d_tiles.alloc((nt + 2) * 8));
      HIP_TRY(launch_wr_count_defined(w.nulls, w.nslots, d_tiles, s));
      HIP_TRY(launch_exclusive_scan(d_tiles, nt, d_tiles.p + nt, s));
      HIP_TRY(hipMemcpyAsync(&w.ndef, d_tiles.p + nt, 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      HIP_TRY(w.dense_owned.alloc(w.ndef * 8));
      HIP_TRY(launch_wr_compact(in[i].values, w.nulls, d_tiles, w.nslots, w.dense_owned, s));
      HIP_TRY(hipStreamSynchronize(s));
      w.dense = w.dense_owned;
    }
    switch (c.storage_type) {
      case ColumnEncoding::UINT64_PLAIN:
      case ColumnEncoding::FLOAT_IEEE754:
      case ColumnEncoding::UINT32_PLAIN:
        break;
      case ColumnEncoding::UINT32_BITPACKED:
      case ColumnEncoding::BOOLEAN_BITPACKED:
        w.bits = bitpack_width(c.bitpack_max_value);
        break;
      case ColumnEncoding::STRING_PLAIN:
      case ColumnEncoding::UINT64_LEB128: {
        const uint64_t nchunks = (w.ndef + kDecodeTile - 1) / kDecodeTile;
        HIP_TRY(w.chunk_offsets.alloc((nchunks + 2) * 8));
        if (nchunks) {
          if (is_string) {
            HIP_TRY(launch_wr_str_count(w.dense, w.ndef, w.chunk_offsets, s));
          } else {
            HIP_TRY(launch_wr_leb_count(w.dense, w.ndef, w.chunk_offsets, s));
          }
          HIP_TRY(launch_exclusive_scan(w.chunk_offsets, nchunks, w.chunk_offsets.p + nchunks, s));
          HIP_TRY(hipMemcpyAsync(&w.leb_bytes, w.chunk_offsets.p + nchunks, 8,
                                 hipMemcpyDeviceToHost, s));
          HIP_TRY(hipStreamSynchronize(s));
        }
        break;
      }
      default:
        return Status::error(EVQL_ENOTSUP, "device writer: column encoding of " + c.name);
    }
  }

  // ---- layout: header, pages, index, metablock ---------------------------------------------
  TableWriter hdr(specs);
  std::vector<uint8_t> head = hdr.image();
  uint64_t pos = head.size();
  struct IndexEntry {
    PageKind kind;
    uint64_t column_id;
    PageRef page;
  };
  std::vector<IndexEntry> index;
  std::unique_ptr<evql_table> t(new evql_table());
  t->ctx = ctx;
  auto bitpacked_pages = [&](PageKind kind, uint64_t cid, uint64_t nvalues, uint32_t bits,
                             std::vector<PageRef>* list) {
    // BitPackedIntPageWriter: zero-width streams write nothing
    if (bits == 0) return;
    const uint64_t nblocks = (nvalues + 127) / 128;
    for (uint64_t blk = 0, pi = 0; blk < nblocks; blk += kBitpackBlocksPerPage, ++pi) {
      const uint32_t sz = 16 * bits * kBitpackBlocksPerPage + (pi == 0 ? 4 : 0);
      list->push_back({pos, sz});
      index.push_back({kind, cid, {pos, sz}});
      pos += sz;
    }
  };
  auto plain_pages = [&](uint64_t cid, uint64_t nbytes, std::vector<PageRef>* list) {
    for (uint64_t b = 0; b < nbytes; b += kPlainPageSize) {
      list->push_back({pos, kPlainPageSize});
      index.push_back({PageKind::DATA, cid, {pos, kPlainPageSize}});
      pos += kPlainPageSize;
    }
  };
  for (size_t i = 0; i < specs.size(); ++i) {
    const ColumnSpec& c = specs[i];
    const ColumnWork& w = work[i];
    ColumnLayout cl;
    cl.name = c.name;
    cl.logical_type = c.logical_type;
    cl.storage_type = c.storage_type;
    cl.column_id = c.column_id;
    cl.rlevel_max = c.rlevel_max;
    cl.dlevel_max = c.dlevel_max;
    uint64_t payload = 0;
    // (ColumnWriter::write* appends the repetition level, then the definition level,
    // then the value: ColumnWriter.cc:59-89)
    if (c.rlevel_max > 0) {
      const uint32_t rbits = bitpack_width(c.rlevel_max);
      bitpacked_pages(PageKind::RLEVEL, c.column_id, w.nslots, rbits, &cl.rlevel_pages);
      if (w.nslots) payload += 4 + 16ull * rbits * ((w.nslots + 127) / 128);
    }
    if (c.dlevel_max > 0) {
      const uint32_t dbits = bitpack_width(c.dlevel_max);
      bitpacked_pages(PageKind::DLEVEL, c.column_id, w.nslots, dbits, &cl.dlevel_pages);
      if (w.nslots) payload += 4 + 16ull * dbits * ((w.nslots + 127) / 128);
    }
    switch (c.storage_type) {
      case ColumnEncoding::UINT64_PLAIN:
      case ColumnEncoding::FLOAT_IEEE754:
        plain_pages(c.column_id, w.ndef * 8, &cl.data_pages);
        payload += w.ndef * 8;
        break;
      case ColumnEncoding::UINT32_PLAIN:
        plain_pages(c.column_id, w.ndef * 4, &cl.data_pages);
        payload += w.ndef * 4;
        break;
      case ColumnEncoding::UINT32_BITPACKED:
      case ColumnEncoding::BOOLEAN_BITPACKED:
        bitpacked_pages(PageKind::DATA, c.column_id, w.ndef, w.bits, &cl.data_pages);
        if (w.ndef && w.bits) payload += 4 + 16ull * w.bits * ((w.ndef + 127) / 128);
        break;
      default:
        plain_pages(c.column_id, w.leb_bytes, &cl.data_pages);
        payload += w.leb_bytes;
    }
    t->layout.columns.push_back(cl);
    t->payload_bytes.push_back(payload);
  }
  const uint64_t index_offset = pos;
  std::vector<uint8_t> idx;
  put_varuint(&idx, index.size());
  for (const auto& e : index) {
    put_varuint(&idx, uint8_t(e.kind));
    put_varuint(&idx, e.column_id);
    put_varuint(&idx, e.page.offset);
    put_varuint(&idx, e.page.size);
  }
  const uint64_t total = index_offset + idx.size();
  {
    // metablock of transaction 1 -> slot 1 (cstable_file.cc:148-184)
    uint8_t mb[kMetaBlockSize];
    memset(mb, 0, sizeof(mb));
    const uint64_t txid = 1;
    memcpy(mb, &txid, 8);
    memcpy(mb + 8, &n, 8);
    memcpy(mb + 16, &index_offset, 8);
    const uint32_t isz = uint32_t(idx.size());
    memcpy(mb + 24, &isz, 4);
    Sha1Digest h = sha1(mb, 28);
    memcpy(mb + 28, h.bytes, 20);
    memcpy(&head[kMetaBlockPosition + kMetaBlockSize * (txid % 2)], mb, kMetaBlockSize);
  }
  t->layout.version = 2;
  t->layout.num_rows = n;
  t->layout.transaction_id = 1;
  t->layout.index_offset = index_offset;
  t->layout.index_size = uint32_t(idx.size());
  t->image_len = total;

  // ---- pass 2: the streams, encoded in place ---------------------------------------------------
  const size_t slack = 1 << 20;  // zero slack behind the image (speculative vector loads)
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t->d_image), total + slack));
  // every page but a stream's last one is overwritten completely by its encoder:
  // zero only the last pages (PageManager pages start zero-filled), the index
  // and the slack
  for (const auto& cl : t->layout.columns) {
    for (const std::vector<PageRef>* list : {&cl.rlevel_pages, &cl.dlevel_pages, &cl.data_pages}) {
      if (list->empty()) continue;
      HIP_TRY(hipMemsetAsync(t->d_image + list->back().offset, 0, list->back().size, s));
    }
  }
  HIP_TRY(hipMemsetAsync(t->d_image + index_offset, 0, total - index_offset + slack, s));
  HIP_TRY(hipMemcpyAsync(t->d_image, head.data(), head.size(), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(t->d_image + index_offset, idx.data(), idx.size(), hipMemcpyHostToDevice,
                         s));
  HIP_TRY(hipStreamSynchronize(s));
  Status st = upload_page_tables(t.get());
  if (!st.ok()) return st;
  for (size_t i = 0; i < specs.size(); ++i) {
    const ColumnSpec& c = specs[i];
    const ColumnWork& w = work[i];
    const ColumnLayout& cl = t->layout.columns[i];
    if (!cl.rlevel_pages.empty()) {
      const uint32_t maxv = c.rlevel_max;
      HIP_TRY(hipMemcpy(t->d_image + cl.rlevel_pages[0].offset, &maxv, 4, hipMemcpyHostToDevice));
      HIP_TRY(launch_wr_bitpack(t->d_image, t->d_pages[i][1], w.rwords, nullptr, w.nslots,
                                bitpack_width(c.rlevel_max), s));
    }
    if (!cl.dlevel_pages.empty()) {
      const uint32_t maxv = c.dlevel_max;
      HIP_TRY(hipMemcpy(t->d_image + cl.dlevel_pages[0].offset, &maxv, 4, hipMemcpyHostToDevice));
      // flat optional columns: the levels are 1 - NULL flag; nested ones: given
      HIP_TRY(launch_wr_bitpack(t->d_image, t->d_pages[i][2], w.dwords.p, w.dwords.p ? nullptr : w.nulls,
                                w.nslots, bitpack_width(c.dlevel_max), s));
    }
    if (cl.data_pages.empty()) continue;
    switch (c.storage_type) {
      case ColumnEncoding::UINT64_PLAIN:
      case ColumnEncoding::FLOAT_IEEE754:
        HIP_TRY(launch_wr_plain(t->d_image, t->d_pages[i][0], w.dense, w.ndef, 8, s));
        break;
      case ColumnEncoding::UINT32_PLAIN:
        HIP_TRY(launch_wr_plain(t->d_image, t->d_pages[i][0], w.dense, w.ndef, 4, s));
        break;
      case ColumnEncoding::UINT32_BITPACKED:
      case ColumnEncoding::BOOLEAN_BITPACKED: {
        const uint32_t maxv = c.bitpack_max_value;
        HIP_TRY(hipMemcpy(t->d_image + cl.data_pages[0].offset, &maxv, 4, hipMemcpyHostToDevice));
        HIP_TRY(launch_wr_bitpack(t->d_image, t->d_pages[i][0], w.dense, nullptr, w.ndef, w.bits, s));
        break;
      }
      case ColumnEncoding::STRING_PLAIN:
        HIP_TRY(launch_wr_str_emit(t->d_image, t->d_pages[i][0], w.dense, w.ndef, w.chunk_offsets,
                                   in[i].bytes, s));
        break;
      default:
        HIP_TRY(launch_wr_leb_emit(t->d_image, t->d_pages[i][0], w.dense, w.ndef, w.chunk_offsets,
                                   s));
    }
  }
  HIP_TRY(hipStreamSynchronize(s));
  *out = t.release();
  return Status();
}

}  // namespace evql
