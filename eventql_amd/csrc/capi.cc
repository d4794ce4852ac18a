// capi.cc -- the extern "C" surface declared in include/evql_gpu.h.
#include <cstring>
#include <fstream>
#include <memory>
#include <chrono>
#include <thread>
#include "runtime.h"
#include "sha1.h"

namespace evql {
const std::string& last_error();
void set_cache_dir(const std::string& d);
Status compile_to_code_object(const std::string& source, std::vector<char>* code, bool use_cache);
Status table_from_image(evql_ctx* ctx, const void* image, size_t len, bool keep_host,
                        evql_table** out);
Status query_prepare(evql_query* q);
Status query_launch(evql_query* q);
Status query_finish(evql_query* q);
Status query_reset(evql_query* q);
Status query_recount(evql_query* q);
Status query_dense_into_table(evql_query* q);
Status chain_merge(evql_query* head);  // exchange.cc
size_t lsm_chain_parts(const evql_lsm_chain* ch, std::vector<evql_table*>* tables,
                       std::vector<const uint8_t*>* d_filters);  // lsm.cc
evql_ctx* lsm_chain_ctx(const evql_lsm_chain* ch);
Status query_reserve_groups(evql_query* q, uint64_t extra);
Status query_import_pairs(evql_query* q, int which, const uint64_t* d_triples, uint64_t n);
Status query_set_order(evql_query* q, const evql_sort_spec_t* specs, uint32_t n, int64_t limit,
                       uint64_t offset);
Status query_next_batch(evql_query* q, size_t max_rows, evql_column_buf_t* cols, size_t* nrows);
}  // namespace evql

using namespace evql;

#define API_TRY try {
#define API_CATCH                                        \
  }                                                      \
  catch (const std::bad_alloc&) {                        \
    return fail(EVQL_ENOMEM, "out of memory");           \
  }                                                      \
  catch (const std::exception& e) {                      \
    return fail(EVQL_ERUNTIME, e.what());                \
  }                                                      \
  catch (...) {                                          \
    return fail(EVQL_ERUNTIME, "unknown error");         \
  }

static int ret(const Status& s) {
  if (s.ok()) return EVQL_OK;
  return fail(s.code, s.msg);
}

template <typename PutFn>
static int writer_put(evql_writer_t* w, int col, uint64_t n, const uint64_t* rlvl,
                      const uint64_t* dlvl, const uint8_t* present, PutFn put) {
  if (col < 0 || size_t(col) >= w->specs.size()) return fail(EVQL_EARG, "bad column index");
  const uint64_t dmax = w->specs[col].dlevel_max;
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t r = rlvl ? rlvl[i] : 0;
    uint64_t d = dlvl ? dlvl[i] : dmax;
    if (present && !present[i]) {
      if (!dlvl) d = dmax > 0 ? dmax - 1 : 0;
      w->w->put_null(col, r, d);
    } else if (d != dmax) {
      w->w->put_null(col, r, d);
    } else {
      put(i, r, d);
    }
  }
  return EVQL_OK;
}


extern "C" {

const char* evql_last_error(void) { return last_error().c_str(); }
const char* evql_version(void) { return "eventql_amd 0.1 (gfx950)"; }

// ---- context -----------------------------------------------------------------------
int evql_ctx_create(int device_ordinal, void* stream, evql_ctx_t** out) {
  API_TRY
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    return fail(EVQL_EDEVICE, "no HIP device available (the MI355X path has no CPU fallback)");
  }
  if (device_ordinal < 0 || device_ordinal >= n) return fail(EVQL_EARG, "bad device ordinal");
  if (hipSetDevice(device_ordinal) != hipSuccess) return fail(EVQL_EDEVICE, "hipSetDevice failed");
  std::unique_ptr<evql_ctx> c(new evql_ctx());
  c->device = device_ordinal;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess) {
    c->num_cus = prop.multiProcessorCount;
  }
  if (stream) {
    c->stream = static_cast<hipStream_t>(stream);
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      return fail(EVQL_EDEVICE, "hipStreamCreate failed");
    }
    c->own_stream = true;
  }
  *out = c.release();
  return EVQL_OK;
  API_CATCH
}

void evql_ctx_destroy(evql_ctx_t* ctx) {
  if (!ctx) return;
  for (auto& kv : ctx->modules) {
    if (kv.second.mod) hipModuleUnload(kv.second.mod);
  }
  if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}

int evql_ctx_synchronize(evql_ctx_t* ctx) {
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(EVQL_EDEVICE, "sync failed");
  return EVQL_OK;
}

void* evql_ctx_stream(evql_ctx_t* ctx) { return ctx->stream; }

// ---- tables -------------------------------------------------------------------------
int evql_table_open_image(evql_ctx_t* ctx, const void* image, size_t len, evql_table_t** out) {
  API_TRY
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(EVQL_EDEVICE, "hipSetDevice failed");
  return ret(table_from_image(ctx, image, len, true, out));
  API_CATCH
}

int evql_table_open_file(evql_ctx_t* ctx, const char* path, evql_table_t** out) {
  API_TRY
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) return fail(EVQL_EIO, std::string("can't open file: ") + path);
  std::streamsize n = f.tellg();
  f.seekg(0);
  std::vector<char> buf((size_t) n);
  if (n > 0 && !f.read(buf.data(), n)) return fail(EVQL_EIO, "read failed");
  return evql_table_open_image(ctx, buf.data(), buf.size(), out);
  API_CATCH
}

void evql_table_close(evql_table_t* t) { delete t; }
uint64_t evql_table_num_rows(const evql_table_t* t) { return t->layout.num_rows; }
int evql_table_num_columns(const evql_table_t* t) { return int(t->layout.columns.size()); }

int evql_table_column_info(const evql_table_t* t, int idx, evql_column_info_t* out) {
  if (idx < 0 || size_t(idx) >= t->layout.columns.size()) return fail(EVQL_EARG, "bad column index");
  const ColumnLayout& c = t->layout.columns[idx];
  memset(out, 0, sizeof(*out));
  strncpy(out->name, c.name.c_str(), sizeof(out->name) - 1);
  out->logical_type = int32_t(c.logical_type);
  out->storage_type = int32_t(c.storage_type);
  out->column_id = c.column_id;
  out->rlevel_max = c.rlevel_max;
  out->dlevel_max = c.dlevel_max;
  out->n_data_pages = uint32_t(c.data_pages.size());
  out->n_rlevel_pages = uint32_t(c.rlevel_pages.size());
  out->n_dlevel_pages = uint32_t(c.dlevel_pages.size());
  out->payload_bytes = t->payload_bytes[idx];
  return EVQL_OK;
}

uint64_t evql_table_image_size(const evql_table_t* t) { return t->image_len; }

int evql_table_download_image(const evql_table_t* t, void* dst, uint64_t len) {
  if (len > t->image_len) len = t->image_len;
  if (hipMemcpy(dst, t->d_image, len, hipMemcpyDeviceToHost) != hipSuccess) {
    return fail(EVQL_EDEVICE, "download failed");
  }
  return EVQL_OK;
}

// xorshift64 step matrix powers M^(2^j) as column images
static void xorshift_jump_matrices(uint64_t (*out)[64], int n) {
  uint64_t m[64];
  for (int bit = 0; bit < 64; ++bit) {
    uint64_t x = 1ull << bit;
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    m[bit] = x;
  }
  for (int j = 0; j < n; ++j) {
    memcpy(out[j], m, sizeof(m));
    uint64_t sq[64];
    for (int bit = 0; bit < 64; ++bit) {
      uint64_t x = m[bit], y = 0;
      for (int b = 0; b < 64; ++b) {
        if ((x >> b) & 1) y ^= m[b];
      }
      sq[bit] = y;
    }
    memcpy(m, sq, sizeof(m));
  }
}

int evql_table_generate(evql_ctx_t* ctx, const evql_synth_spec_t* spec, evql_table_t** out) {
  API_TRY
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(EVQL_EDEVICE, "hipSetDevice failed");
  const uint64_t n = spec->num_rows;
  struct Gen {
    const char* name;
    uint32_t bit;
    ColumnType lt;
    ColumnEncoding enc;
  };
  const Gen gens[] = {{"k", 1, ColumnType::UNSIGNED_INT, ColumnEncoding::UINT64_PLAIN},
                      {"a", 2, ColumnType::UNSIGNED_INT, ColumnEncoding::UINT64_PLAIN},
                      {"b", 4, ColumnType::UNSIGNED_INT, ColumnEncoding::UINT64_PLAIN},
                      {"v", 8, ColumnType::FLOAT, ColumnEncoding::FLOAT_IEEE754},
                      {"u", 16, ColumnType::UNSIGNED_INT, ColumnEncoding::UINT64_PLAIN}};
  // header through the regular writer (no values), pages laid out by hand exactly
  // as PageManager would allocate them for a column-at-a-time load
  std::vector<ColumnSpec> specs;
  uint64_t cid = 0;
  for (const auto& g : gens) {
    if (!(spec->columns & g.bit)) continue;
    ColumnSpec cs{g.name, g.lt, g.enc, ++cid, 0, 0, 0xffffffffu};
    if (g.bit == 1 && spec->k_bits > 0) {
      cs.storage_type = ColumnEncoding::UINT32_BITPACKED;
      cs.bitpack_max_value = spec->k_bits >= 32 ? 0xffffffffu : ((1u << spec->k_bits) - 1);
    }
    specs.push_back(cs);
  }
  TableWriter hdr(specs);
  uint64_t pos = hdr.image().size();
  struct IdxE {
    uint64_t cid, off;
    uint32_t size;
  };
  std::vector<IdxE> index;
  std::unique_ptr<SynthArgs> sa(new SynthArgs());
  memset(sa.get(), 0, sizeof(SynthArgs));
  for (const auto& cs : specs) {
    uint64_t first = pos;
    if (cs.storage_type == ColumnEncoding::UINT32_BITPACKED) {
      const uint32_t b = bitpack_width(cs.bitpack_max_value);
      const uint64_t nblocks = (n + 127) / 128;
      for (uint64_t blk = 0, pi = 0; blk < nblocks; blk += kBitpackBlocksPerPage, ++pi) {
        uint32_t sz = 16 * b * kBitpackBlocksPerPage + (pi == 0 ? 4 : 0);
        index.push_back({cs.column_id, pos, sz});
        pos += sz;
      }
    } else {
      const uint64_t npages = (n * 8 + kPlainPageSize - 1) / kPlainPageSize;
      for (uint64_t p = 0; p < npages; ++p) {
        index.push_back({cs.column_id, pos, kPlainPageSize});
        pos += kPlainPageSize;
      }
    }
    if (cs.name == "k") sa->off_k = first;
    if (cs.name == "a") sa->off_a = first;
    if (cs.name == "b") sa->off_b = first;
    if (cs.name == "v") sa->off_v = first;
    if (cs.name == "u") sa->off_u = first;
  }
  const uint64_t index_offset = pos;
  std::vector<uint8_t> idx;
  auto varuint = [&](uint64_t v) {
    do {
      uint8_t x = v & 0x7f;
      v >>= 7;
      if (v) x |= 0x80;
      idx.push_back(x);
    } while (v);
  };
  varuint(index.size());
  for (const auto& e : index) {
    varuint(1);
    varuint(e.cid);
    varuint(e.off);
    varuint(e.size);
  }
  const uint64_t total = index_offset + idx.size();

  // header with the metablock: reuse the writer's commit on an empty table and
  // patch the index location
  std::vector<uint8_t> head = hdr.image();
  {
    uint8_t mb[48];
    memset(mb, 0, sizeof(mb));
    const uint64_t txid = 1;
    memcpy(mb, &txid, 8);
    memcpy(mb + 8, &n, 8);
    memcpy(mb + 16, &index_offset, 8);
    uint32_t isz = uint32_t(idx.size());
    memcpy(mb + 24, &isz, 4);
    Sha1Digest h = sha1(mb, 28);
    memcpy(mb + 28, h.bytes, 20);
    memcpy(&head[kMetaBlockPosition + kMetaBlockSize * (txid % 2)], mb, 48);
  }

  std::unique_ptr<evql_table> t(new evql_table());
  t->ctx = ctx;
  t->image_len = total;
  const size_t slack = 1 << 20;
  if (hipMalloc(reinterpret_cast<void**>(&t->d_image), total + slack) != hipSuccess) {
    return fail(EVQL_ENOMEM, "hipMalloc of the table image failed");
  }
  hipStream_t s = ctx->stream;
  hipMemsetAsync(t->d_image, 0, total + slack, s);
  hipMemcpyAsync(t->d_image, head.data(), head.size(), hipMemcpyHostToDevice, s);
  hipMemcpyAsync(t->d_image + index_offset, idx.data(), idx.size(), hipMemcpyHostToDevice, s);
  for (const auto& cs : specs) {
    if (cs.storage_type == ColumnEncoding::UINT32_BITPACKED) {
      uint64_t off = sa->off_k;
      hipMemcpyAsync(t->d_image + off, &cs.bitpack_max_value, 4, hipMemcpyHostToDevice, s);
    }
  }
  hipStreamSynchronize(s);
  sa->image = t->d_image;
  sa->num_rows = n;
  sa->seed = spec->seed;
  sa->k_mod = spec->k_mod ? spec->k_mod : 1000;
  sa->u_mod = spec->u_mod ? spec->u_mod : 1;
  sa->columns = spec->columns;
  sa->k_bits = spec->k_bits;
  xorshift_jump_matrices(sa->jump, 48);
  SynthArgs* d_args = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d_args), sizeof(SynthArgs)) != hipSuccess) {
    return fail(EVQL_ENOMEM, "hipMalloc failed");
  }
  hipMemcpy(d_args, sa.get(), sizeof(SynthArgs), hipMemcpyHostToDevice);
  hipError_t le = launch_synth(d_args, n, s);
  hipError_t se = hipStreamSynchronize(s);
  hipFree(d_args);
  if (le != hipSuccess || se != hipSuccess) return fail(EVQL_EDEVICE, "synthetic table kernel failed");

  // layout straight from what was laid out (no need to read the image back)
  t->layout.version = 2;
  t->layout.num_rows = n;
  t->layout.transaction_id = 1;
  t->layout.index_offset = index_offset;
  t->layout.index_size = uint32_t(idx.size());
  for (const auto& cs : specs) {
    ColumnLayout cl;
    cl.name = cs.name;
    cl.logical_type = cs.logical_type;
    cl.storage_type = cs.storage_type;
    cl.column_id = cs.column_id;
    cl.rlevel_max = cl.dlevel_max = 0;
    for (const auto& e : index) {
      if (e.cid == cs.column_id) cl.data_pages.push_back({e.off, e.size});
    }
    t->layout.columns.push_back(cl);
    uint64_t payload = cs.storage_type == ColumnEncoding::UINT32_BITPACKED
                           ? 4 + 16ull * bitpack_width(cs.bitpack_max_value) * ((n + 127) / 128)
                           : 8 * n;
    t->payload_bytes.push_back(payload);
  }
  // device page tables
  t->d_pages.assign(t->layout.columns.size(), std::vector<uint64_t*>(3, nullptr));
  for (size_t i = 0; i < t->layout.columns.size(); ++i) {
    for (int k = 0; k < 3; ++k) {
      std::vector<uint64_t> offs;
      if (k == 0) {
        for (const auto& p : t->layout.columns[i].data_pages) offs.push_back(p.offset);
      }
      if (offs.empty()) offs.push_back(0);
      offs.push_back(offs.back());
      uint64_t* d = nullptr;
      if (hipMalloc(reinterpret_cast<void**>(&d), offs.size() * 8) != hipSuccess) {
        return fail(EVQL_ENOMEM, "hipMalloc failed");
      }
      hipMemcpy(d, offs.data(), offs.size() * 8, hipMemcpyHostToDevice);
      t->d_pages[i][k] = d;
    }
  }
  *out = t.release();
  return EVQL_OK;
  API_CATCH
}

// ---- device-side writer -----------------------------------------------------------------
int evql_table_from_device_columns(evql_ctx_t* ctx, const evql_column_spec_t* cols, int ncols,
                                   const evql_device_column_t* data, uint64_t num_rows,
                                   evql_table_t** out) {
  return evql_table_from_device_columns_ordered(ctx, cols, ncols, data, num_rows,
                                                EVQL_PAGE_ORDER_COLUMNS, out);
}

int evql_table_from_device_columns_ordered(evql_ctx_t* ctx, const evql_column_spec_t* cols,
                                           int ncols, const evql_device_column_t* data,
                                           uint64_t num_rows, int page_order,
                                           evql_table_t** out) {
  API_TRY
  if (!ctx || !cols || !data || !out || ncols <= 0) return fail(EVQL_EARG, "bad arguments");
  if (page_order != EVQL_PAGE_ORDER_COLUMNS && page_order != EVQL_PAGE_ORDER_ROWS) {
    return fail(EVQL_EARG, "bad page order");
  }
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(EVQL_EDEVICE, "hipSetDevice failed");
  std::vector<ColumnSpec> specs;
  std::vector<DeviceColumnIn> in;
  for (int i = 0; i < ncols; ++i) {
    ColumnSpec cs;
    cs.name = cols[i].name;
    cs.logical_type = ColumnType(cols[i].logical_type);
    cs.storage_type = ColumnEncoding(cols[i].storage_type);
    cs.column_id = cols[i].column_id;
    cs.rlevel_max = cols[i].rlevel_max;
    cs.dlevel_max = cols[i].dlevel_max;
    cs.bitpack_max_value = cols[i].bitpack_max_value;
    if (cs.bitpack_max_value == 0) {
      cs.bitpack_max_value =
          cs.storage_type == ColumnEncoding::BOOLEAN_BITPACKED ? 1u : 0xffffffffu;
    }
    specs.push_back(cs);
    in.push_back({data[i].values, data[i].nulls, data[i].bytes, data[i].rlevels, data[i].dlevels,
                  data[i].num_slots});
  }
  evql_table* t = nullptr;
  Status st = table_from_device_columns(ctx, specs, in, num_rows, page_order, &t);
  if (!st.ok()) return ret(st);
  *out = t;
  return EVQL_OK;
  API_CATCH
}

// ---- writer --------------------------------------------------------------------------
int evql_writer_create(const evql_column_spec_t* cols, int ncols, evql_writer_t** out) {
  API_TRY
  std::unique_ptr<evql_writer> w(new evql_writer());
  for (int i = 0; i < ncols; ++i) {
    ColumnSpec cs;
    cs.name = cols[i].name;
    cs.logical_type = ColumnType(cols[i].logical_type);
    cs.storage_type = ColumnEncoding(cols[i].storage_type);
    cs.column_id = cols[i].column_id;
    cs.rlevel_max = cols[i].rlevel_max;
    cs.dlevel_max = cols[i].dlevel_max;
    cs.bitpack_max_value = cols[i].bitpack_max_value;
    if (cs.bitpack_max_value == 0) {
      cs.bitpack_max_value =
          cs.storage_type == ColumnEncoding::BOOLEAN_BITPACKED ? 1u : 0xffffffffu;
    }
    w->specs.push_back(cs);
  }
  w->w.reset(new TableWriter(w->specs));
  *out = w.release();
  return EVQL_OK;
  API_CATCH
}

int evql_writer_put_uint(evql_writer_t* w, int col, uint64_t n, const uint64_t* rlvl,
                         const uint64_t* dlvl, const uint8_t* present, const uint64_t* values) {
  API_TRY
  return writer_put(w, col, n, rlvl, dlvl, present,
                    [&](uint64_t i, uint64_t r, uint64_t d) { w->w->put_uint(col, r, d, values[i]); });
  API_CATCH
}

int evql_writer_put_float(evql_writer_t* w, int col, uint64_t n, const uint64_t* rlvl,
                          const uint64_t* dlvl, const uint8_t* present, const double* values) {
  API_TRY
  return writer_put(w, col, n, rlvl, dlvl, present,
                    [&](uint64_t i, uint64_t r, uint64_t d) { w->w->put_float(col, r, d, values[i]); });
  API_CATCH
}

int evql_writer_put_string(evql_writer_t* w, int col, uint64_t n, const uint64_t* rlvl,
                           const uint64_t* dlvl, const uint8_t* present, const uint64_t* offsets,
                           const char* bytes) {
  API_TRY
  return writer_put(w, col, n, rlvl, dlvl, present, [&](uint64_t i, uint64_t r, uint64_t d) {
    w->w->put_string(col, r, d, bytes + offsets[i], offsets[i + 1] - offsets[i]);
  });
  API_CATCH
}

int evql_writer_commit(evql_writer_t* w, uint64_t num_rows) {
  API_TRY
  w->w->commit(num_rows);
  return EVQL_OK;
  API_CATCH
}

const void* evql_writer_image(const evql_writer_t* w, uint64_t* len) {
  *len = w->w->image().size();
  return w->w->image().data();
}

int evql_cstable_upgrade(const void* image, uint64_t len, void* dst, uint64_t dst_cap,
                         uint64_t* out_len) {
  if (!image || !out_len) return fail(EVQL_EARG, "null argument");
  std::vector<uint8_t> v2;
  std::string e = transcode_v1_to_v2(static_cast<const uint8_t*>(image), len, &v2);
  if (!e.empty()) return fail(EVQL_EIO, e);
  *out_len = v2.size();
  if (!dst || dst_cap < v2.size()) return fail(EVQL_EARG, "destination too small");
  memcpy(dst, v2.data(), v2.size());
  return EVQL_OK;
}

int evql_cstable_inspect(const void* image, uint64_t len, uint64_t* num_rows, int* num_columns) {
  API_TRY
  if (!image) return fail(EVQL_EARG, "null argument");
  const uint8_t* b = static_cast<const uint8_t*>(image);
  std::vector<uint8_t> v2;
  if (len >= 6 && b[0] == 0x23 && b[1] == 0x17 && b[2] == 0x23 && b[3] == 0x17 &&
      (uint32_t(b[4]) | (uint32_t(b[5]) << 8)) == 1) {
    std::string e = transcode_v1_to_v2(b, len, &v2);
    if (!e.empty()) return fail(EVQL_EIO, e);
    b = v2.data();
    len = v2.size();
  }
  TableLayout layout;
  std::string e = parse_cstable(b, len, &layout);
  if (!e.empty()) return fail(EVQL_EIO, e);
  if (num_rows) *num_rows = layout.num_rows;
  if (num_columns) *num_columns = int(layout.columns.size());
  return EVQL_OK;
  API_CATCH
}

int evql_writer_write_file(const evql_writer_t* w, const char* path) {
  std::string e = w->w->write_file(path);
  if (!e.empty()) return fail(EVQL_EIO, e);
  return EVQL_OK;
}

void evql_writer_destroy(evql_writer_t* w) { delete w; }

// ---- queries --------------------------------------------------------------------------
// one operator over one table.  d_filter != nullptr: a row filter that already sits in HBM
// (an evql_lsm_chain's; borrowed) instead of plan->row_filter_bits
static int create_one(evql_ctx_t* ctx, evql_table_t* table, const evql_plan_desc_t* plan_in,
                      const uint8_t* d_filter, evql_query_t** out) {
  evql_plan_desc_t plan_copy = *plan_in;
  const evql_plan_desc_t* plan = &plan_copy;
  std::unique_ptr<evql_query> q(new evql_query());
  q->ctx = ctx;
  q->table = table;
  q->group_mode = plan->group_mode;
  q->groups_hint = plan->groups_hint;
  q->float_sum_mode = plan->float_sum_mode;
  q->float_sum_bound = plan->float_sum_bound;
  if (plan->float_sum_mode > EVQL_FLOAT_SUM_EXACT || !(plan->float_sum_bound >= 0)) {
    return fail(EVQL_EARG, "bad float sum mode / bound");
  }
  q->row_begin = plan->row_begin;
  q->row_end = plan->row_end;
  if (plan->group_mode != EVQL_MODE_FINAL && plan->group_mode != EVQL_MODE_PARTIAL) {
    return fail(EVQL_EARG, "bad group mode");
  }
  if (d_filter) {
    static const uint8_t marker = 0;  // (the planner only asks whether there is a filter)
    plan_copy.row_filter_bits = &marker;
    plan_copy.row_filter_len = table->layout.num_rows;
    q->row_filter_len = table->layout.num_rows;
    q->d_row_filter = const_cast<uint8_t*>(d_filter);
    q->row_filter_owned = false;
  } else if (plan->row_filter_bits) {
    q->row_filter_len = plan->row_filter_len;
    q->row_filter_host.assign(plan->row_filter_bits,
                              plan->row_filter_bits + (plan->row_filter_len + 7) / 8);
  }
  bool unsupported = false;
  Status st = build_kernel_plan(table->layout, plan, q.get(), &unsupported);
  if (!st.ok()) return ret(st);
  st = query_prepare(q.get());
  if (!st.ok()) return ret(st);
  *out = q.release();
  return EVQL_OK;
}

int evql_query_create(evql_ctx_t* ctx, evql_table_t* table, const evql_plan_desc_t* plan,
                      evql_query_t** out) {
  API_TRY
  if (!ctx || !table || !plan || !out) return fail(EVQL_EARG, "null argument");
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(EVQL_EDEVICE, "hipSetDevice failed");
  return create_one(ctx, table, plan, nullptr, out);
  API_CATCH
}

int evql_query_create_chain(evql_ctx_t* ctx, evql_lsm_chain_t* ch, const evql_plan_desc_t* plan,
                            evql_query_t** out) {
  API_TRY
  if (!ctx || !ch || !plan || !out) return fail(EVQL_EARG, "null argument");
  if (lsm_chain_ctx(ch) != ctx) return fail(EVQL_EARG, "chain belongs to another context");
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(EVQL_EDEVICE, "hipSetDevice failed");
  std::vector<evql_table*> tables;
  std::vector<const uint8_t*> d_filters;
  if (lsm_chain_parts(ch, &tables, &d_filters) == 0) return fail(EVQL_EARG, "chain was not built");
  if (plan->row_filter_bits || plan->row_begin || plan->row_end) {
    return fail(EVQL_EARG, "a chain scan takes its row filters from the chain");
  }
  if (tables.size() > 1 && plan->scan_mode != EVQL_SCAN_FLAT) {
    // PartitionCursor builds CSTableScan for aggregating statements (partition_cursor.cc:
    // 205-213); its setFilter is not lowered
    return fail(EVQL_ENOTSUP, "nested scan over a chain of tables");
  }
  if (tables.size() > 1 && plan->n_select == 0 && plan->n_group == 0) {
    return fail(EVQL_ENOTSUP, "bare scan over a chain of tables");
  }
  std::vector<std::unique_ptr<evql_query>> parts;
  for (size_t i = 0; i < tables.size(); ++i) {
    // a table without a filter is created as a plain scan (setFilter is not called)
    evql_query* q = nullptr;
    int rc = create_one(ctx, tables[i], plan, d_filters[i], &q);
    if (rc != EVQL_OK) return rc;
    parts.emplace_back(q);
  }
  evql_query* head = parts[0].get();
  // exact float sums: one quantum for the whole chain (the words of the parts are added)
  for (int k = 0; k < 4; ++k) {
    int ex = head->fsum_exp[k];
    double b = head->fsum_bound[k];
    for (auto& p : parts) {
      ex = std::max(ex, p->fsum_exp[k]);
      b = std::max(b, p->fsum_bound[k]);
    }
    for (auto& p : parts) {
      p->fsum_exp[k] = ex;
      p->fsum_bound[k] = b;
    }
  }
  for (size_t i = 1; i < parts.size(); ++i) head->chain.push_back(parts[i].release());
  *out = parts[0].release();
  return EVQL_OK;
  API_CATCH
}

void evql_query_destroy(evql_query_t* q) { delete q; }

// a chain query runs the scans of all its tables (head first), then merges their groups
static Status launch_all(evql_query* q) {
  Status st = query_launch(q);
  for (size_t i = 0; st.ok() && i < q->chain.size(); ++i) st = query_launch(q->chain[i]);
  return st;
}

static Status finish_all(evql_query* q) {
  Status st = query_finish(q);
  for (size_t i = 0; st.ok() && i < q->chain.size(); ++i) st = query_finish(q->chain[i]);
  if (st.ok() && !q->chain.empty()) st = chain_merge(q);
  return st;
}

int evql_query_launch(evql_query_t* q) {
  API_TRY
  return ret(launch_all(q));
  API_CATCH
}

int evql_query_finish(evql_query_t* q) {
  API_TRY
  return ret(finish_all(q));
  API_CATCH
}

int evql_query_execute(evql_query_t* q, evql_heartbeat_fn hb, void* user) {
  API_TRY
  if (hb && hb(user) != 0) return fail(EVQL_ERUNTIME, "query aborted by heartbeat");
  // (the cardinality probe of a hint-less plan and the re-runs after a full table happen
  // inside launch / finish: they beat through the query)
  struct HbScope {
    std::vector<evql_query*> qs;
    ~HbScope() {
      for (evql_query* p : qs) {
        p->hb = nullptr;
        p->hb_user = nullptr;
      }
    }
  } hbs;
  hbs.qs.push_back(q);
  for (evql_query* c : q->chain) hbs.qs.push_back(c);
  for (evql_query* p : hbs.qs) {
    p->hb = hb;
    p->hb_user = user;
    p->hb_abort = false;
  }
  Status st = launch_all(q);
  if (!st.ok()) return ret(st);
  // GroupByExpression::execute calls txn_->triggerHeartbeat() once per input batch
  // (groupby.cc:100-105) so that a long scan keeps its connection alive; here the scan
  // is a kernel in flight: the host polls the stream and beats every 5 ms meanwhile.
  // A heartbeat that asks to stop is honoured once the kernels have drained (a running
  // grid is not cancelled).
  bool aborted = false;
  if (hb) {
    auto last = std::chrono::steady_clock::now();
    while (hipStreamQuery(q->ctx->stream) == hipErrorNotReady) {
      std::this_thread::sleep_for(std::chrono::microseconds(200));
      const auto now = std::chrono::steady_clock::now();
      if (now - last >= std::chrono::milliseconds(5)) {
        last = now;
        if (hb(user) != 0) aborted = true;
      }
    }
  }
  st = finish_all(q);
  if (!st.ok()) return ret(st);
  for (evql_query* p : hbs.qs) aborted = aborted || p->hb_abort;
  if (aborted || (hb && hb(user) != 0)) return fail(EVQL_ERUNTIME, "query aborted by heartbeat");
  return EVQL_OK;
  API_CATCH
}

int evql_query_column_count(const evql_query_t* q) {
  // PartialGroupByExpression emits (key, data) string pairs (groupby.cc:484-491)
  if (q->group_mode == EVQL_MODE_PARTIAL) return 2;
  return int(q->select.size());
}
int evql_query_column_type(const evql_query_t* q, int idx) {
  if (q->group_mode == EVQL_MODE_PARTIAL) return idx >= 0 && idx < 2 ? EVQL_T_STRING : -1;
  if (idx < 0 || size_t(idx) >= q->select.size()) return -1;
  return int(q->select[idx].return_type);
}

int evql_query_next_batch(evql_query_t* q, size_t max_rows, evql_column_buf_t* cols,
                          size_t* nrows) {
  API_TRY
  return ret(query_next_batch(q, max_rows, cols, nrows));
  API_CATCH
}

int evql_query_stats(const evql_query_t* q, evql_query_stats_t* out) {
  *out = q->stats;
  uint64_t bytes = 0;
  // (a dictionary-coded key counts as the string column it stands for)
  for (const auto& c : q->within_record ? q->wr_cols : q->rplan().cols) {
    bytes += q->table->payload_bytes[c.layout_index];
  }
  // scaled to the scanned row range; + result bytes (key + 8 B per aggregate)
  const uint64_t nrows = q->table->layout.num_rows;
  if (nrows && q->stats.rows_scanned != nrows && q->chain.empty()) {
    bytes = uint64_t(double(bytes) * double(q->stats.rows_scanned) / double(nrows));
  }
  for (const evql_query* part : q->chain) {  // (chain_merge summed the row counters)
    for (const auto& c : part->rplan().cols) bytes += part->table->payload_bytes[c.layout_index];
  }
  bytes += q->stats.num_groups * 8 * (1 + q->kp.aggs.size());
  out->algorithmic_bytes = bytes;
  return EVQL_OK;
}

const char* evql_query_kernel_source(const evql_query_t* q) { return q->source.c_str(); }

// ---- partial aggregates ---------------------------------------------------------------
uint32_t evql_query_record_words(const evql_query_t* q) {
  return uint32_t(q->rplan().words_per_slot()) + 1;
}

int evql_query_partial_view(evql_query_t* q, evql_partial_view_t* out) {
  if (!q || !out) return fail(EVQL_EARG, "null argument");
  if (q->merged) return fail(EVQL_EARG, "the query's groups were merged (exchange / chain): emit them with next_batch");
  if (q->dict_key) return fail(EVQL_ENOTSUP, "plan groups by dictionary codes: merge it with evql_query_exchange");
  {
    Status st = query_dense_into_table(q);
    if (!st.ok()) return ret(st);
  }
  out->device_words = q->d_gtab;
  out->capacity = q->gcap + 8;
  out->words_per_group = uint32_t(q->kp.words_per_slot());
  out->num_groups = q->ngroups;
  return EVQL_OK;
}

int evql_query_export_groups(evql_query_t* q, void* device_dst, uint64_t max_groups,
                             uint64_t* n_groups) {
  API_TRY
  if (q->merged) return fail(EVQL_EARG, "the query's groups were merged (exchange / chain): emit them with next_batch");
  if (q->kp.n_exact > 0 && !(q->float_sum_bound > 0)) {
    // the records carry integer multiples of a quantum derived from ONE table's maxima;
    // another partition may have chosen a different one
    return fail(EVQL_EARG, "exact float sums travel between partitions only with an explicit float_sum_bound");
  }
  if (q->rplan().need_first_row) {
    // a first-row index means something only inside the table that produced it:
    // the key / select values of such plans travel inside the records of
    // evql_query_exchange (exchange.cc, k_resolve_records)
    return fail(EVQL_ENOTSUP, "plan reads first-row values: merge it with evql_query_exchange");
  }
  hipStream_t s = q->ctx->stream;
  if (!q->d_gtab || !q->d_counters) return fail(EVQL_EARG, "execute() was not called");
  uint64_t* d_cnt = q->d_counters + 6;  // per-query counter block: no allocation per call
  const uint32_t nwords = uint32_t(q->kp.words_per_slot());
  // (partitioned path) the dense records as they are, the table's groups behind them
  const uint64_t nd = q->dense_n;
  if (nd > max_groups) return fail(EVQL_ENOMEM, "export buffer too small");
  uint64_t* dst = static_cast<uint64_t*>(device_dst);
  if (nd) hipMemcpyAsync(dst, q->d_dense, nd * (nwords + 1) * 8, hipMemcpyDeviceToDevice, s);
  hipMemsetAsync(d_cnt, 0, 8, s);
  hipError_t e = hipSuccess;
  uint64_t n = 0;
  if (q->ngroups > nd) {
    e = launch_table_compact(q->d_gtab, q->gcap, q->gcap + 8, nwords, dst + nd * (nwords + 1),
                             max_groups - nd, d_cnt, s);
    hipMemcpyAsync(&n, d_cnt, 8, hipMemcpyDeviceToHost, s);
  }
  hipError_t e2 = hipStreamSynchronize(s);
  if (e != hipSuccess || e2 != hipSuccess) return fail(EVQL_EDEVICE, "export kernel failed");
  if (n > max_groups - nd) return fail(EVQL_ENOMEM, "export buffer too small");
  *n_groups = n + nd;
  return EVQL_OK;
  API_CATCH
}

int evql_query_import_groups(evql_query_t* q, const void* device_src, uint64_t n_groups) {
  API_TRY
  if (q->merged) return fail(EVQL_EARG, "the query's groups were merged (exchange / chain): emit them with next_batch");
  if (q->kp.n_exact > 0 && !(q->float_sum_bound > 0)) {
    // the records carry integer multiples of a quantum derived from ONE table's maxima;
    // another partition may have chosen a different one
    return fail(EVQL_EARG, "exact float sums travel between partitions only with an explicit float_sum_bound");
  }
  if (q->rplan().need_first_row) {
    return fail(EVQL_ENOTSUP, "plan reads first-row values: merge it with evql_query_exchange");
  }
  hipStream_t s = q->ctx->stream;
  {
    // (also moves the dense records of the partitioned path into the table)
    Status st = query_reserve_groups(q, n_groups);
    if (!st.ok()) return ret(st);
  }
  MergeArgs a{};
  a.words = q->d_gtab;
  a.gcap = q->gcap;
  a.stride = q->gcap + 8;
  a.nwords = uint32_t(q->kp.words_per_slot());
  int w = 1;
  a.has_ident2 = q->kp.has_ident2() ? 1 : 0;
  if (q->kp.has_ident2()) a.ops[w++] = 255;
  if (q->kp.need_first_row) a.ops[w++] = 2;  // min
  for (const auto& sw : q->kp.states) a.ops[w++] = uint32_t(sw.op);
  // a count_distinct word counts the pairs of THIS query's set: foreign counts are not
  // added, the pairs arrive through evql_query_import_pairs and are counted there
  for (const auto& ag : q->kp.aggs) {
    if (ag.distinct_index >= 0) a.ops[q->kp.state_word_base() + ag.first_word] = 254;  // no such op: skipped
  }
  a.status = q->d_status;
  hipMemsetAsync(q->d_status, 0, 16, s);
  hipError_t e = launch_table_merge(a, static_cast<const uint64_t*>(device_src), n_groups, s);
  uint32_t status[4] = {0};
  hipMemcpyAsync(status, q->d_status, 16, hipMemcpyDeviceToHost, s);
  if (e != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
    return fail(EVQL_EDEVICE, "merge kernel failed");
  }
  if (status[0] & 2u) {
    // cannot happen for capacity (reserved above); a probe chain beyond the bound would
    // leave the merge half done, so the aggregate is unusable from here on
    q->executed = false;
    q->ngroups = 0;
    return fail(EVQL_ENOMEM, "group table probe bound exceeded during merge; the query must be re-run");
  }
  // the host copy of the result (if any) is stale now
  return ret(query_recount(q));
  API_CATCH
}

uint32_t evql_query_distinct_aggregates(const evql_query_t* q) { return q ? uint32_t(q->kp.n_distinct) : 0; }

int evql_query_export_pairs(evql_query_t* q, uint32_t which, void* device_dst, uint64_t max_pairs,
                            uint64_t* n_pairs) {
  API_TRY
  if (!q || !n_pairs) return fail(EVQL_EARG, "null argument");
  if (which >= uint32_t(q->kp.n_distinct)) return fail(EVQL_EARG, "no such count_distinct aggregate");
  if (q->merged) return fail(EVQL_EARG, "the query's groups were merged (exchange / chain): emit them with next_batch");
  hipStream_t s = q->ctx->stream;
  *n_pairs = 0;
  if (!q->d_pairset[which]) return EVQL_OK;  // (nothing scanned yet: an empty merge target)
  uint64_t* d_cnt = q->d_counters + 6;
  hipMemsetAsync(d_cnt, 0, 8, s);
  hipError_t e = launch_pairset_export(q->d_pairset[which], q->pairset_cap,
                                       static_cast<uint64_t*>(device_dst), device_dst ? max_pairs : 0,
                                       d_cnt, s);
  uint64_t n = 0;
  hipMemcpyAsync(&n, d_cnt, 8, hipMemcpyDeviceToHost, s);
  if (e != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return fail(EVQL_EDEVICE, "export kernel failed");
  *n_pairs = n;  // (device_dst == NULL: the count only)
  if (device_dst && n > max_pairs) return fail(EVQL_ENOMEM, "export buffer too small");
  return EVQL_OK;
  API_CATCH
}

int evql_query_import_pairs(evql_query_t* q, uint32_t which, const void* device_src,
                            uint64_t n_pairs) {
  API_TRY
  if (!q || (!device_src && n_pairs)) return fail(EVQL_EARG, "null argument");
  if (which >= uint32_t(q->kp.n_distinct)) return fail(EVQL_EARG, "no such count_distinct aggregate");
  if (q->merged) return fail(EVQL_EARG, "the query's groups were merged (exchange / chain): emit them with next_batch");
  if (!q->d_gtab) return fail(EVQL_EARG, "import the group records first");
  return ret(query_import_pairs(q, int(which), static_cast<const uint64_t*>(device_src), n_pairs));
  API_CATCH
}

int evql_query_reset(evql_query_t* q) {
  API_TRY
  if (q->dict_key) return fail(EVQL_ENOTSUP, "plan groups by dictionary codes: not a merge target");
  return ret(query_reset(q));
  API_CATCH
}

int evql_query_set_order(evql_query_t* q, const evql_sort_spec_t* specs, uint32_t n_specs,
                         int64_t limit, uint64_t offset) {
  API_TRY
  if (!q || (n_specs && !specs)) return fail(EVQL_EARG, "null argument");
  return ret(query_set_order(q, specs, n_specs, limit, offset));
  API_CATCH
}

// ---- build support ------------------------------------------------------------------------
void evql_set_kernel_cache_dir(const char* dir) { set_cache_dir(dir ? dir : ""); }

int evql_compile_only(const evql_plan_desc_t* plan, const evql_column_info_t* columns, int ncolumns,
                      const char* cache_dir, size_t* code_size) {
  API_TRY
  if (cache_dir) set_cache_dir(cache_dir);
  TableLayout layout;
  layout.version = 2;
  layout.num_rows = 0;
  for (int i = 0; i < ncolumns; ++i) {
    ColumnLayout c;
    c.name = columns[i].name;
    c.logical_type = ColumnType(columns[i].logical_type);
    c.storage_type = ColumnEncoding(columns[i].storage_type);
    c.column_id = columns[i].column_id;
    c.rlevel_max = columns[i].rlevel_max;
    c.dlevel_max = columns[i].dlevel_max;
    layout.columns.push_back(c);
  }
  evql_query q;
  bool unsupported = false;
  Status st = build_kernel_plan(layout, plan, &q, &unsupported);
  if (!st.ok()) return ret(st);
  // bit widths are not known without pages: payload_bytes carries the width
  for (auto& c : q.kp.cols) {
    if (c.mode == ColAccess::BITPACKED) c.bits = uint32_t(columns[c.layout_index].payload_bytes);
  }
  q.source = generate_kernel_source(q.kp);
  std::vector<char> code;
  st = compile_to_code_object(q.source, &code, true);
  if (!st.ok()) return ret(st);
  if (code_size) *code_size = code.size();
  return EVQL_OK;
  API_CATCH
}

}  // extern "C"
