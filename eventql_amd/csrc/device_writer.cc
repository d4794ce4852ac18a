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
// at the running file offset in the order the page writers first need them, so the
// byte layout of a reference-written file depends on the order of the write calls.
// Both orders a caller of the reference can produce are reproduced byte for byte:
//   EVQL_PAGE_ORDER_COLUMNS  one whole column after the other (what cstable_format.cc's
//                            TableWriter and a ColumnWriter-per-column loop produce)
//   EVQL_PAGE_ORDER_ROWS     row by row, every column per row in schema order
//                            (RecordShredder::addRecord*, cstable_writer.cc addRow loops)
// A page is allocated by the write that first does not fit: level page k (bit-packed,
// 131,072 values) at slot 131,072 k; a fixed-width data page at its first value; a
// LEB128 / string page by the value whose encoding holds stream byte 524,288 k.  Those
// value indexes are found on the device (k_wr_select_byte), mapped to the slot / record
// that writes them (k_wr_select_zero over the NULL flags, k_wr_rank_zero over the
// repetition levels) and the allocations sorted on the host -- a few thousand entries.
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
#include <algorithm>
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
  DevBuf<uint64_t> def_tiles;       // scanned per-tile counts of the defined slots
};

// one PageManager::allocPage call of the sequential writer
struct PageAlloc {
  uint64_t record;  // row / record whose write allocates the page (ROWS order)
  uint64_t slot;    // slot of that column (value position incl. NULLs)
  uint32_t column;
  uint32_t stream;  // 0 repetition levels, 1 definition levels, 2 data (the order of
                    // ColumnWriter::write*, ColumnWriter.cc:59-89)
  uint32_t page;    // index within the stream
  uint32_t size;
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
                                 int page_order, evql_table** out) {
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
      DevBuf<uint64_t>& d_tiles = w.def_tiles;
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
  // every page allocation of the sequential writer, then their order
  std::vector<PageAlloc> allocs;
  for (size_t i = 0; i < specs.size(); ++i) {
    const ColumnSpec& c = specs[i];
    ColumnWork& w = work[i];
    auto level_pages = [&](uint32_t stream, uint32_t bits) {
      // BitPackedIntPageWriter (page_writer_bitpacked.cc:43-67): zero-width streams
      // write nothing; page k is allocated by value 131,072 k
      if (bits == 0) return;
      const uint64_t nblocks = (w.nslots + 127) / 128;
      for (uint64_t blk = 0, pi = 0; blk < nblocks; blk += kBitpackBlocksPerPage, ++pi) {
        const uint32_t sz = 16 * bits * kBitpackBlocksPerPage + (pi == 0 ? 4 : 0);
        allocs.push_back({0, blk * 128, uint32_t(i), stream, uint32_t(pi), sz});
      }
    };
    if (c.rlevel_max > 0) level_pages(0, bitpack_width(c.rlevel_max));
    if (c.dlevel_max > 0) level_pages(1, bitpack_width(c.dlevel_max));
    // data pages: index of the (defined) value that allocates each
    std::vector<uint64_t> first_value;
    std::vector<uint32_t> sizes;
    switch (c.storage_type) {
      case ColumnEncoding::UINT64_PLAIN:
      case ColumnEncoding::FLOAT_IEEE754:
        for (uint64_t v = 0; v < w.ndef; v += kPlainPageSize / 8) {
          first_value.push_back(v);
          sizes.push_back(kPlainPageSize);
        }
        break;
      case ColumnEncoding::UINT32_PLAIN:
        for (uint64_t v = 0; v < w.ndef; v += kPlainPageSize / 4) {
          first_value.push_back(v);
          sizes.push_back(kPlainPageSize);
        }
        break;
      case ColumnEncoding::UINT32_BITPACKED:
      case ColumnEncoding::BOOLEAN_BITPACKED:
        if (w.bits) {
          for (uint64_t v = 0, pi = 0; v < w.ndef; v += 128ull * kBitpackBlocksPerPage, ++pi) {
            first_value.push_back(v);
            sizes.push_back(16 * w.bits * kBitpackBlocksPerPage + (pi == 0 ? 4 : 0));
          }
        }
        break;
      default: {  // LEB128 / strings: the value holding stream byte 524,288 k
        const uint64_t npages = (w.leb_bytes + kPlainPageSize - 1) / kPlainPageSize;
        if (npages) {
          std::vector<uint64_t> qs(npages);
          for (uint64_t k = 0; k < npages; ++k) qs[k] = k * uint64_t(kPlainPageSize);
          DevBuf<uint64_t> d_q;
          HIP_TRY(d_q.alloc(npages * 16));
          HIP_TRY(hipMemcpyAsync(d_q, qs.data(), npages * 8, hipMemcpyHostToDevice, s));
          HIP_TRY(launch_wr_select_byte(w.dense, w.ndef, w.chunk_offsets,
                                        c.storage_type == ColumnEncoding::STRING_PLAIN, d_q, npages,
                                        d_q.p + npages, s));
          first_value.resize(npages);
          HIP_TRY(hipMemcpyAsync(first_value.data(), d_q.p + npages, npages * 8,
                                 hipMemcpyDeviceToHost, s));
          HIP_TRY(hipStreamSynchronize(s));
          sizes.assign(npages, kPlainPageSize);
        }
      }
    }
    // ... and the slot that carries it (NULL / undefined slots carry no value)
    std::vector<uint64_t> slot_of = first_value;
    if (c.dlevel_max > 0 && !first_value.empty()) {
      const uint64_t nq = first_value.size();
      DevBuf<uint64_t> d_q;
      HIP_TRY(d_q.alloc(nq * 16));
      HIP_TRY(hipMemcpyAsync(d_q, first_value.data(), nq * 8, hipMemcpyHostToDevice, s));
      HIP_TRY(launch_wr_select_zero(w.nulls, w.nslots, w.def_tiles, d_q, nq, d_q.p + nq, s));
      HIP_TRY(hipMemcpyAsync(slot_of.data(), d_q.p + nq, nq * 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
    }
    for (size_t k = 0; k < first_value.size(); ++k) {
      allocs.push_back({0, slot_of[k], uint32_t(i), 2, uint32_t(k), sizes[k]});
    }
  }
  if (page_order == EVQL_PAGE_ORDER_ROWS) {
    // the record that writes each slot: flat columns one slot per row; repeated columns
    // the number of slots with repetition level 0 up to and including it, minus one
    for (size_t i = 0; i < specs.size(); ++i) {
      std::vector<size_t> mine;
      for (size_t a = 0; a < allocs.size(); ++a) {
        if (allocs[a].column == i) mine.push_back(a);
      }
      if (specs[i].rlevel_max == 0 || mine.empty()) {
        for (size_t a : mine) allocs[a].record = allocs[a].slot;
        continue;
      }
      const ColumnWork& w = work[i];
      const uint64_t nt = (w.nslots + kDecodeTile - 1) / kDecodeTile;
      DevBuf<uint64_t> d_tiles, d_q;
      HIP_TRY(d_tiles.alloc((nt + 2) * 8));
      HIP_TRY(launch_wr_count_defined(in[i].rlevels, w.nslots, d_tiles, s));  // counts zeros
      HIP_TRY(launch_exclusive_scan(d_tiles, nt, d_tiles.p + nt, s));
      std::vector<uint64_t> qs(mine.size()), rk(mine.size());
      for (size_t k = 0; k < mine.size(); ++k) qs[k] = allocs[mine[k]].slot;
      HIP_TRY(d_q.alloc(qs.size() * 16));
      HIP_TRY(hipMemcpyAsync(d_q, qs.data(), qs.size() * 8, hipMemcpyHostToDevice, s));
      HIP_TRY(launch_wr_rank_zero(in[i].rlevels, w.nslots, d_tiles, d_q, qs.size(),
                                  d_q.p + qs.size(), s));
      HIP_TRY(hipMemcpyAsync(rk.data(), d_q.p + qs.size(), qs.size() * 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      for (size_t k = 0; k < mine.size(); ++k) allocs[mine[k]].record = rk[k] ? rk[k] - 1 : 0;
    }
    std::stable_sort(allocs.begin(), allocs.end(), [](const PageAlloc& a, const PageAlloc& b) {
      if (a.record != b.record) return a.record < b.record;
      if (a.column != b.column) return a.column < b.column;
      if (a.slot != b.slot) return a.slot < b.slot;
      return a.stream < b.stream;
    });
  } else {
    std::stable_sort(allocs.begin(), allocs.end(), [](const PageAlloc& a, const PageAlloc& b) {
      if (a.column != b.column) return a.column < b.column;
      if (a.slot != b.slot) return a.slot < b.slot;
      return a.stream < b.stream;
    });
  }
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
    if (c.rlevel_max > 0 && w.nslots) {
      payload += 4 + 16ull * bitpack_width(c.rlevel_max) * ((w.nslots + 127) / 128);
    }
    if (c.dlevel_max > 0 && w.nslots) {
      payload += 4 + 16ull * bitpack_width(c.dlevel_max) * ((w.nslots + 127) / 128);
    }
    switch (c.storage_type) {
      case ColumnEncoding::UINT64_PLAIN:
      case ColumnEncoding::FLOAT_IEEE754:
        payload += w.ndef * 8;
        break;
      case ColumnEncoding::UINT32_PLAIN:
        payload += w.ndef * 4;
        break;
      case ColumnEncoding::UINT32_BITPACKED:
      case ColumnEncoding::BOOLEAN_BITPACKED:
        if (w.ndef && w.bits) payload += 4 + 16ull * w.bits * ((w.ndef + 127) / 128);
        break;
      default:
        payload += w.leb_bytes;
    }
    t->layout.columns.push_back(cl);
    t->payload_bytes.push_back(payload);
  }
  for (const PageAlloc& a : allocs) {
    ColumnLayout& cl = t->layout.columns[a.column];
    std::vector<PageRef>& list =
        a.stream == 0 ? cl.rlevel_pages : (a.stream == 1 ? cl.dlevel_pages : cl.data_pages);
    // (within one stream the allocations are already in page order)
    list.push_back({pos, a.size});
    index.push_back({a.stream == 0 ? PageKind::RLEVEL : (a.stream == 1 ? PageKind::DLEVEL : PageKind::DATA),
                     specs[a.column].column_id, {pos, a.size}});
    pos += a.size;
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
