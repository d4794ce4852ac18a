// runtime.cc -- HIP device context, table residency in HBM, JIT of the fused
// kernel, execution and result emission.
#include "runtime.h"
#include <dlfcn.h>
#include <unistd.h>
#include <atomic>
#include <algorithm>
#include <cmath>
#include <limits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <sys/stat.h>
#include "sha1.h"

namespace evql {

static thread_local std::string g_last_error;
static std::string g_cache_dir;
static bool g_cache_dir_set = false;

// Default place of the on-disk kernel cache: `_kcache` next to this shared library (the
// directory the python package points evql_set_kernel_cache_dir at too), so that every
// host of the library -- the adapter inside evqld, the probe, python -- shares the
// compiled plan kernels.  evql_set_kernel_cache_dir("") switches the disk cache off.
static const std::string& cache_dir() {
  if (!g_cache_dir_set) {
    g_cache_dir_set = true;
    Dl_info info;
    if (dladdr(reinterpret_cast<const void*>(&cache_dir), &info) && info.dli_fname) {
      std::string path = info.dli_fname;
      const size_t slash = path.rfind('/');
      g_cache_dir = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/_kcache";
    }
  }
  return g_cache_dir;
}

void set_last_error(const std::string& m) { g_last_error = m; }
const std::string& last_error() { return g_last_error; }
int fail(int code, const std::string& m) {
  g_last_error = m;
  return code;
}
void set_cache_dir(const std::string& d) {
  g_cache_dir = d;
  g_cache_dir_set = true;
}

#define HIP_TRY(expr)                                                             \
  do {                                                                            \
    hipError_t e__ = (expr);                                                      \
    if (e__ != hipSuccess) {                                                      \
      return Status::error(EVQL_EDEVICE, std::string(#expr) + ": " +              \
                                             hipGetErrorString(e__));             \
    }                                                                             \
  } while (0)

static std::string hex_digest(const std::string& s) {
  Sha1Digest d = sha1(s.data(), s.size());
  char b[41];
  for (int i = 0; i < 20; ++i) snprintf(b + 2 * i, 3, "%02x", d.bytes[i]);
  return std::string(b, 40);
}

// ---------------------------------------------------------------------------
// kernel compilation (hiprtc, gfx950) with an in-memory and an on-disk cache
// ---------------------------------------------------------------------------
static const char* kCompileOptions[] = {"--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics",
                                        "-ffp-contract=off", "-std=c++17"};

// an ELF header and, where it can be checked cheaply, a section header table inside
// the file: what a complete code object of the cache looks like
static bool plausible_code_object(const std::vector<char>& c) {
  if (c.size() < 64 || memcmp(c.data(), "\x7f" "ELF", 4) != 0) return false;
  uint64_t shoff;
  uint16_t shentsize, shnum;
  memcpy(&shoff, c.data() + 0x28, 8);
  memcpy(&shentsize, c.data() + 0x3a, 2);
  memcpy(&shnum, c.data() + 0x3c, 2);
  return shoff <= c.size() && uint64_t(shentsize) * shnum <= c.size() - shoff;
}

Status compile_to_code_object(const std::string& source, std::vector<char>* code, bool use_cache) {
  const std::string full = std::string(device_library_source()) + "\n" + source;
  std::string key = hex_digest(full);
  std::string cache_file;
  const std::string& dir = cache_dir();
  if (!dir.empty()) {
    cache_file = dir + "/" + key + ".hsaco";
    std::ifstream f(cache_file, std::ios::binary);
    if (f && use_cache) {
      code->assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
      if (plausible_code_object(*code)) return Status();
      code->clear();
    }
  }
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, full.c_str(), "evql_fused.hip", 0, nullptr, nullptr) !=
      HIPRTC_SUCCESS) {
    return Status::error(EVQL_EDEVICE, "hiprtcCreateProgram failed");
  }
  hiprtcResult rc = hiprtcCompileProgram(
      prog, int(sizeof(kCompileOptions) / sizeof(kCompileOptions[0])), kCompileOptions);
  if (rc != HIPRTC_SUCCESS) {
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    std::string log(ls, '\0');
    if (ls) hiprtcGetProgramLog(prog, &log[0]);
    hiprtcDestroyProgram(&prog);
    return Status::error(EVQL_EDEVICE, "kernel compilation failed: " + log);
  }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  code->resize(cs);
  hiprtcGetCode(prog, code->data());
  hiprtcDestroyProgram(&prog);
  if (!cache_file.empty()) {
    mkdir(dir.c_str(), 0755);
    // (several processes -- one per GPU -- compile the same plan at the same time: each
    // writes a file of its own and renames it into place)
    static std::atomic<unsigned> serial{0};
    const std::string tmp = cache_file + "." + std::to_string(long(getpid())) + "." +
                            std::to_string(serial.fetch_add(1)) + ".tmp";
    std::ofstream f(tmp, std::ios::binary);
    f.write(code->data(), std::streamsize(code->size()));
    f.close();
    if (!f || rename(tmp.c_str(), cache_file.c_str()) != 0) remove(tmp.c_str());
  }
  return Status();
}

Status compile_kernel(evql_ctx* ctx, const std::string& source, Module* out, bool load_module) {
  const std::string key = hex_digest(source);
  if (ctx) {
    auto it = ctx->modules.find(key);
    if (it != ctx->modules.end()) {
      *out = it->second;
      return Status();
    }
  }
  std::vector<char> code;
  Status st = compile_to_code_object(source, &code, true);
  if (!st.ok()) return st;
  out->code_size = code.size();
  if (load_module) {
    if (hipModuleLoadData(&out->mod, code.data()) != hipSuccess) {
      // a damaged cache file: compile again (and replace it)
      (void) hipGetLastError();
      st = compile_to_code_object(source, &code, false);
      if (!st.ok()) return st;
      out->code_size = code.size();
      HIP_TRY(hipModuleLoadData(&out->mod, code.data()));
    }
    HIP_TRY(hipModuleGetFunction(&out->fn, out->mod, "evql_scan_agg"));
    if (source.find("evql_part_aggregate") != std::string::npos) {
      out->fn_count = nullptr;  // (absent from the fused form, KernelPlan::part_fused)
      if (source.find("evql_part_count(") != std::string::npos) {
        HIP_TRY(hipModuleGetFunction(&out->fn_count, out->mod, "evql_part_count"));
      }
      HIP_TRY(hipModuleGetFunction(&out->fn_scatter, out->mod, "evql_part_scatter"));
      HIP_TRY(hipModuleGetFunction(&out->fn_aggregate, out->mod, "evql_part_aggregate"));
      if (source.find("evql_part_refine") != std::string::npos) {
        HIP_TRY(hipModuleGetFunction(&out->fn_refine, out->mod, "evql_part_refine"));
      }
    }
    if (source.find("evql_where_rows") != std::string::npos) {
      HIP_TRY(hipModuleGetFunction(&out->fn_where, out->mod, "evql_where_rows"));
    }
    if (ctx) ctx->modules[key] = *out;
  }
  return Status();
}

// ---------------------------------------------------------------------------
// tables
// ---------------------------------------------------------------------------
static uint64_t stream_payload_bytes(const std::vector<uint8_t>& img, const ColumnLayout& c,
                                     uint64_t nrows) {
  // encoded bytes actually holding values
  uint64_t total = 0;
  auto bitpacked = [&](const std::vector<PageRef>& pages) -> uint64_t {
    if (pages.empty()) return 0;
    uint32_t maxv;
    memcpy(&maxv, &img[pages[0].offset], 4);
    return 4 + 16ull * bitpack_width(maxv) * ((nrows + 127) / 128);
  };
  auto bytes_used = [&](const std::vector<PageRef>& pages) -> uint64_t {
    if (pages.empty()) return 0;
    uint64_t full = 0;
    for (size_t i = 0; i + 1 < pages.size(); ++i) full += pages[i].size;
    const PageRef& last = pages.back();
    uint64_t used = last.size;
    while (used > 0 && img[last.offset + used - 1] == 0) --used;
    return full + used;
  };
  switch (c.storage_type) {
    case ColumnEncoding::UINT64_PLAIN:
    case ColumnEncoding::FLOAT_IEEE754:
      total += c.dlevel_max == 0 ? 8 * nrows : bytes_used(c.data_pages);
      break;
    case ColumnEncoding::UINT32_PLAIN:
      total += c.dlevel_max == 0 ? 4 * nrows : bytes_used(c.data_pages);
      break;
    case ColumnEncoding::UINT32_BITPACKED:
    case ColumnEncoding::BOOLEAN_BITPACKED:
      total += c.dlevel_max == 0 ? bitpacked(c.data_pages) : bytes_used(c.data_pages);
      break;
    default:
      total += bytes_used(c.data_pages);
  }
  if (c.dlevel_max > 0) total += bitpacked(c.dlevel_pages);
  if (c.rlevel_max > 0) total += bitpacked(c.rlevel_pages);
  return total;
}

Status upload_page_tables(evql_table* t) {
  t->d_pages.assign(t->layout.columns.size(), std::vector<uint64_t*>(3, nullptr));
  for (size_t i = 0; i < t->layout.columns.size(); ++i) {
    const ColumnLayout& c = t->layout.columns[i];
    const std::vector<PageRef>* lists[3] = {&c.data_pages, &c.rlevel_pages, &c.dlevel_pages};
    for (int k = 0; k < 3; ++k) {
      std::vector<uint64_t> offs;
      for (const auto& p : *lists[k]) offs.push_back(p.offset);
      if (offs.empty()) offs.push_back(0);
      // one extra entry so that a tile index one past the end stays in bounds
      offs.push_back(offs.back());
      uint64_t* d = nullptr;
      HIP_TRY(hipMalloc(&d, offs.size() * 8));
      HIP_TRY(hipMemcpyAsync(d, offs.data(), offs.size() * 8, hipMemcpyHostToDevice,
                             t->ctx->stream));
      HIP_TRY(hipStreamSynchronize(t->ctx->stream));
      t->d_pages[i][k] = d;
    }
  }
  return Status();
}

Status table_from_image(evql_ctx* ctx, const void* image, size_t len, bool keep_host,
                        evql_table** out) {
  std::unique_ptr<evql_table> t(new evql_table());
  t->ctx = ctx;
  std::vector<uint8_t> transcoded;
  {
    // v0.1.0 files are re-encoded into the v0.2.0 page layout first (cstable_v1.cc)
    const uint8_t* b = static_cast<const uint8_t*>(image);
    if (len >= 6 && b[0] == 0x23 && b[1] == 0x17 && b[2] == 0x23 && b[3] == 0x17 &&
        (uint32_t(b[4]) | (uint32_t(b[5]) << 8)) == 1) {
      std::string verr = transcode_v1_to_v2(b, len, &transcoded);
      if (!verr.empty()) return Status::error(EVQL_EIO, verr);
      image = transcoded.data();
      len = transcoded.size();
    }
  }
  std::string err = parse_cstable(static_cast<const uint8_t*>(image), len, &t->layout);
  if (!err.empty()) return Status::error(EVQL_EIO, err);
  t->image_len = len;
  const uint8_t* img = static_cast<const uint8_t*>(image);
  std::vector<uint8_t> tmp(img, img + len);
  for (const auto& c : t->layout.columns) {
    t->payload_bytes.push_back(stream_payload_bytes(tmp, c, t->layout.num_rows));
  }
  // 1 MiB of zero slack behind the image keeps speculative vector loads legal
  const size_t slack = 1 << 20;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t->d_image), len + slack));
  HIP_TRY(hipMemsetAsync(t->d_image + len, 0, slack, ctx->stream));
  HIP_TRY(hipMemcpyAsync(t->d_image, image, len, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  (void) keep_host;  // nothing of the file is kept on the host
  Status st = upload_page_tables(t.get());
  if (!st.ok()) return st;
  *out = t.release();
  return Status();
}

}  // namespace evql

evql_table::~evql_table() {
  if (d_image) hipFree(d_image);
  for (auto& v : d_pages) {
    for (auto* p : v) {
      if (p) hipFree(p);
    }
  }
  // (`materialized` columns free their device arrays themselves)
  for (auto& kv : nested_cache) {
    if (kv.second.d_values) hipFree(kv.second.d_values);
    if (kv.second.d_hash) hipFree(kv.second.d_hash);
    if (kv.second.d_packed) hipFree(kv.second.d_packed);
    if (kv.second.d_packed_pages) hipFree(kv.second.d_packed_pages);
  }
  for (auto& kv : leaf_cache) {
    if (kv.second.levels) hipFree(kv.second.levels);
    if (kv.second.rec_offsets) hipFree(kv.second.rec_offsets);
  }
}

evql_query::~evql_query() {
  if (d_gtab) hipFree(d_gtab);
  if (d_status) hipFree(d_status);
  if (d_counters) hipFree(d_counters);
  if (d_small_rec) hipFree(d_small_rec);
  for (auto* p : d_pairset) {
    if (p) hipFree(p);
  }
  for (auto* p : d_mset) {
    if (p) hipFree(p);
  }
  if (d_row_filter && row_filter_owned) hipFree(d_row_filter);
  for (auto* c : chain) delete c;
  if (d_part_counts) hipFree(d_part_counts);
  if (d_bucket_start) hipFree(d_bucket_start);
  if (d_tuples) hipFree(d_tuples);
  if (d_tuples_tmp) hipFree(d_tuples_tmp);
  if (d_part_cursors) hipFree(d_part_cursors);
  if (d_dense) hipFree(d_dense);
  if (d_mtab) hipFree(d_mtab);
  if (d_mdense) hipFree(d_mdense);
  if (d_conv) hipFree(d_conv);
  for (auto* p : demit.col) {
    if (p) hipHostFree(p);
  }
  for (auto* p : demit.off) {
    if (p) hipHostFree(p);
  }
  for (auto* p : nested_owned) hipFree(p);
  if (ev0) hipEventDestroy(ev0);
  if (ev1) hipEventDestroy(ev1);
}

namespace evql {

// ---------------------------------------------------------------------------
// decode-to-SoA ("materialise") of columns the fused kernel cannot read in place
// ---------------------------------------------------------------------------
static uint64_t padded_rows(uint64_t n) {
  const uint64_t pad = 8192;  // largest tile
  return (n + pad - 1) / pad * pad + pad;
}

static Status stream_bits(evql_table* t, const std::vector<PageRef>& pages, uint32_t* bits);

// Value boundaries of a STRING_PLAIN column, on the device (aot_kernels.h
// "STRING_PLAIN value boundaries"): d_strval[i] = (len << 40) | position of value i's
// first byte in the virtual byte stream over the column's 512 KiB data pages.
static Status locate_string_values(evql_table* t, const ColumnLayout& c, int li, uint64_t nvalues,
                                   uint64_t* d_strval) {
  hipStream_t s = t->ctx->stream;
  if (nvalues == 0) return Status();
  StrScanArgs a{};
  a.image = t->d_image;
  a.pages = t->d_pages[li][0];
  a.nbytes = uint64_t(c.data_pages.size()) * kPlainPageSize;
  a.nchunks = a.nbytes / kStrChunk;
  a.nvalues = nvalues;
  a.strval = d_strval;
  if (a.nchunks == 0) return Status::error(EVQL_EIO, "end of column reached: " + c.name);
  const uint64_t ngroups = (a.nchunks + kStrGroup - 1) / kStrGroup;
  DevBuf<uint16_t> d_exits, d_hops, d_centry;
  DevBuf<uint32_t> d_gexit, d_ghops, d_status;
  DevBuf<uint64_t> d_gentry, d_gbase, d_cbase;
  HIP_TRY(d_exits.alloc(a.nchunks * kStrEntries * 2));
  HIP_TRY(d_hops.alloc(a.nchunks * kStrEntries * 2));
  HIP_TRY(d_gexit.alloc(ngroups * kStrEntries * 4));
  HIP_TRY(d_ghops.alloc(ngroups * kStrEntries * 4));
  HIP_TRY(d_gentry.alloc(ngroups * 8));
  HIP_TRY(d_gbase.alloc(ngroups * 8));
  HIP_TRY(d_centry.alloc(a.nchunks * 2));
  HIP_TRY(d_cbase.alloc(a.nchunks * 8));
  HIP_TRY(d_status.alloc(16));
  HIP_TRY(hipMemsetAsync(d_status, 0, 16, s));
  a.exits = d_exits;
  a.hops = d_hops;
  a.gexit = d_gexit;
  a.ghops = d_ghops;
  a.gentry = d_gentry;
  a.gbase = d_gbase;
  a.centry = d_centry;
  a.cbase = d_cbase;
  a.status = d_status;
  HIP_TRY(launch_str_chunk_tables(a, s));
  HIP_TRY(launch_str_group_compose(a, s));
  HIP_TRY(launch_str_chain(a, s));
  HIP_TRY(launch_str_chunk_entries(a, s));
  uint32_t status[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(status, d_status, 16, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (status[0] & 1u) {
    // a value outran the chunk tables: locate the chunk entries with the serial walk
    HIP_TRY(hipMemsetAsync(d_status, 0, 16, s));
    HIP_TRY(launch_str_walk_serial(a, s));
    HIP_TRY(hipMemcpyAsync(status, d_status, 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  uint64_t found;
  memcpy(&found, &status[2], 8);
  if ((status[0] & 2u) || found < nvalues) {
    return Status::error(EVQL_EIO, "end of column reached: " + c.name);
  }
  HIP_TRY(launch_str_emit(a, s));
  HIP_TRY(hipMemcpyAsync(status, d_status, 16, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (status[0] & 2u) return Status::error(EVQL_EIO, "end of column reached: " + c.name);
  return Status();
}

// values the data pages of a fixed-width encoding can hold (byte streams: no bound,
// their decoders stop at the end of the stream)
static uint64_t fixed_width_capacity(const ColumnLayout& c) {
  switch (c.storage_type) {
    case ColumnEncoding::UINT64_PLAIN:
    case ColumnEncoding::FLOAT_IEEE754:
      return uint64_t(c.data_pages.size()) * (kPlainPageSize / 8);
    case ColumnEncoding::UINT32_PLAIN:
      return uint64_t(c.data_pages.size()) * (kPlainPageSize / 4);
    case ColumnEncoding::UINT32_BITPACKED:
    case ColumnEncoding::BOOLEAN_BITPACKED:
      return c.data_pages.empty() ? ~0ull
                                  : uint64_t(c.data_pages.size()) * kBitpackBlocksPerPage * 128;
    default:
      return ~0ull;
  }
}

// `n` u64 values as bit-packed pages (libsimdcomp layout, 131,072 values per page) of the
// narrowest of 8 / 16 / 32 bits that holds their maximum; *bits = 0 when it does not fit
// 32 bits.  Widths dividing 32 never straddle a word: the decode is one shift and one mask.
static Status pack_narrow(hipStream_t s, const uint64_t* d_values, uint64_t n, uint8_t** d_packed,
                          uint64_t** d_packed_pages, uint32_t* bits_out) {
  *bits_out = 0;
  if (n == 0) return Status();
  DevBuf<uint64_t> d_max;
  HIP_TRY(d_max.alloc(8));
  HIP_TRY(hipMemsetAsync(d_max, 0, 8, s));
  HIP_TRY(launch_max_u64(d_values, n, d_max, s));
  uint64_t maxv = 0;
  HIP_TRY(hipMemcpyAsync(&maxv, d_max, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (maxv > 0xffffffffull) return Status();
  const uint32_t bits = maxv <= 0xffu ? 8 : (maxv <= 0xffffu ? 16 : 32);
  const uint64_t nblocks = (n + 127) / 128;
  const uint64_t page_bytes = 16ull * bits * kBitpackBlocksPerPage;
  const uint64_t npages = (nblocks + kBitpackBlocksPerPage - 1) / kBitpackBlocksPerPage;
  // a tile reads up to 8192 rows beyond the last one: zero slack like the image's
  const uint64_t bytes = 4 + npages * page_bytes + (1 << 20);
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(d_packed), bytes));
  HIP_TRY(hipMemsetAsync(*d_packed, 0, bytes, s));
  std::vector<uint64_t> offs;
  for (uint64_t pi = 0; pi < npages; ++pi) offs.push_back(pi == 0 ? 0 : 4 + pi * page_bytes);
  offs.push_back(offs.back());  // (one past the end stays in bounds)
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(d_packed_pages), offs.size() * 8));
  HIP_TRY(hipMemcpyAsync(*d_packed_pages, offs.data(), offs.size() * 8, hipMemcpyHostToDevice, s));
  const uint32_t hdr = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
  HIP_TRY(hipMemcpyAsync(*d_packed, &hdr, 4, hipMemcpyHostToDevice, s));
  HIP_TRY(launch_wr_bitpack(*d_packed, *d_packed_pages, d_values, nullptr, n, bits, s));
  HIP_TRY(hipStreamSynchronize(s));
  *bits_out = bits;
  return Status();
}

static Status materialize_column(evql_table* t, const ColAccess& ca, uint32_t* bits_out) {
  evql_ctx* ctx = t->ctx;
  const ColumnLayout& c = t->layout.columns[ca.layout_index];
  (void) bits_out;
  if (t->materialized.count(c.name)) return Status();
  MaterializedColumn m;
  const uint64_t n = t->layout.num_rows;
  const uint64_t np = padded_rows(n);
  hipStream_t s = ctx->stream;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m.d_values), np * 8));
  HIP_TRY(hipMemsetAsync(m.d_values, 0, np * 8, s));
  const int li = ca.layout_index;

  if (c.logical_type == ColumnType::STRING) {
    // per row: tag, (len << 40 | position) and a 64-bit hash of the bytes -- all
    // computed on the device from the pages in HBM (no host copy of the file)
    m.string_hash = true;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m.d_tags), np));
    HIP_TRY(hipMemsetAsync(m.d_tags, 0, np, s));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m.d_strpos), np * 8));
    HIP_TRY(hipMemsetAsync(m.d_strpos, 0, np * 8, s));
    uint64_t nvalues = n;
    DevBuf<uint64_t> d_tiles;
    const uint64_t ntiles = (n + kDecodeTile - 1) / kDecodeTile;
    if (c.dlevel_max > 0) {
      HIP_TRY(hipMemsetAsync(m.d_tags, 1, np, s));
      HIP_TRY(d_tiles.alloc((ntiles + 1) * 8));
      uint32_t dbits = 0;
      Status st = stream_bits(t, c.dlevel_pages, &dbits);
      if (!st.ok()) return st;
      HIP_TRY(launch_dlevel_tags(t->d_image, t->d_pages[li][2], dbits, c.dlevel_max, n, m.d_tags,
                                 d_tiles, s));
      uint64_t* d_total = d_tiles.p + ntiles;
      HIP_TRY(launch_exclusive_scan(d_tiles, ntiles, d_total, s));
      HIP_TRY(hipMemcpyAsync(&nvalues, d_total, 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      DevBuf<uint64_t> d_strval;
      HIP_TRY(d_strval.alloc(nvalues * 8));
      Status st2 = locate_string_values(t, c, li, nvalues, d_strval);
      if (!st2.ok()) return st2;
      RtColumn src{};
      src.mode = ColAccess::SOA;
      src.soa = d_strval;
      HIP_TRY(launch_expand_nullable(t->d_image, src, m.d_tags, d_tiles, n, m.d_strpos, s));
      HIP_TRY(hipStreamSynchronize(s));
    } else {
      Status st2 = locate_string_values(t, c, li, n, m.d_strpos);
      if (!st2.ok()) return st2;
    }
    HIP_TRY(launch_string_hash(t->d_image, t->d_pages[li][0], m.d_strpos, n, m.d_values, s));
    HIP_TRY(hipStreamSynchronize(s));
    t->materialized[c.name] = std::move(m);
    return Status();
  }

  // where do defined values come from?
  RtColumn src{};
  src.pages = t->d_pages[li][0];
  DevBuf<uint64_t> d_dense;  // LEB128 decoded (nullable columns only)
  uint64_t nvalues = n;
  DevBuf<uint64_t> d_tiles;
  const uint64_t ntiles = (n + kDecodeTile - 1) / kDecodeTile;

  if (c.dlevel_max > 0) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m.d_tags), np));
    HIP_TRY(hipMemsetAsync(m.d_tags, 1, np, s));
    HIP_TRY(d_tiles.alloc((ntiles + 1) * 8));
    uint32_t maxv = 0;
    if (!c.dlevel_pages.empty()) {
      HIP_TRY(hipMemcpy(&maxv, t->d_image + c.dlevel_pages[0].offset, 4, hipMemcpyDeviceToHost));
    }
    const uint32_t dbits = c.dlevel_pages.empty() ? 0 : bitpack_width(maxv);
    HIP_TRY(launch_dlevel_tags(t->d_image, t->d_pages[li][2], dbits, c.dlevel_max, n, m.d_tags,
                               d_tiles, s));
    uint64_t* d_total = d_tiles.p + ntiles;
    HIP_TRY(launch_exclusive_scan(d_tiles, ntiles, d_total, s));
    HIP_TRY(hipMemcpyAsync(&nvalues, d_total, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }

  if (nvalues > fixed_width_capacity(c)) {
    // fewer data pages than defined values (column_reader_uint.cc raises
    // "end of column reached" when it gets there)
    return Status::error(EVQL_EIO, "end of column reached: " + c.name);
  }
  switch (c.storage_type) {
    case ColumnEncoding::UINT64_PLAIN:
    case ColumnEncoding::FLOAT_IEEE754:
      src.mode = ColAccess::PLAIN64;
      break;
    case ColumnEncoding::UINT32_PLAIN:
      src.mode = ColAccess::PLAIN32;
      break;
    case ColumnEncoding::UINT32_BITPACKED:
    case ColumnEncoding::BOOLEAN_BITPACKED: {
      src.mode = ColAccess::BITPACKED;
      uint32_t maxv = 0;
      if (!c.data_pages.empty()) {
        HIP_TRY(hipMemcpy(&maxv, t->d_image + c.data_pages[0].offset, 4, hipMemcpyDeviceToHost));
      }
      src.bits = c.data_pages.empty() ? 0 : bitpack_width(maxv);
      break;
    }
    case ColumnEncoding::UINT64_LEB128: {
      const uint64_t nbytes = uint64_t(c.data_pages.size()) * kPlainPageSize;
      const uint64_t nchunks = (nbytes + kLebChunk - 1) / kLebChunk;
      DevBuf<uint64_t> d_chunks;
      HIP_TRY(d_chunks.alloc((nchunks + 1) * 8));
      // a non-nullable column decodes straight into its SoA array; a nullable one
      // into a dense array that the expansion below reads by value index
      uint64_t* dst = m.d_values;
      if (c.dlevel_max > 0) {
        HIP_TRY(d_dense.alloc(nvalues * 8));
        dst = d_dense;
      }
      if (nchunks) {
        HIP_TRY(launch_leb128_count(t->d_image, t->d_pages[li][0], nbytes, d_chunks, s));
        HIP_TRY(launch_exclusive_scan(d_chunks, nchunks, nullptr, s));
        HIP_TRY(launch_leb128_decode(t->d_image, t->d_pages[li][0], nbytes, d_chunks, nvalues, dst,
                                     s));
      }
      HIP_TRY(hipStreamSynchronize(s));
      src.mode = ColAccess::SOA;
      src.soa = dst;
      break;
    }
    default:
      return Status::error(EVQL_ENOTSUP, "unsupported column encoding");
  }

  if (c.dlevel_max > 0) {
    HIP_TRY(launch_expand_nullable(t->d_image, src, m.d_tags, d_tiles, n, m.d_values, s));
    HIP_TRY(hipStreamSynchronize(s));
  } else if (c.storage_type == ColumnEncoding::UINT64_LEB128 && n > 0) {
    // Required LEB128 column (the reference's default integer encoding,
    // TableSchema.cc:290-316): keep it in HBM as bit-packed pages of the narrowest
    // of 8 / 16 / 32 bits that holds its maximum instead of 8-byte words.  The fused
    // kernel then streams fewer bytes than the LEB128 stream itself holds for
    // multi-byte values, with a decode of one shift and one mask (widths dividing 32
    // never straddle a word).  Decoding LEB128 inside the fused kernel instead would
    // cost ~10 lane-operations per stream byte (terminator scan + extraction) against
    // the ~12 the chip has per HBM byte at 6.3 TB/s for the whole query.
    uint32_t bits = 0;
    Status stp = pack_narrow(s, m.d_values, n, &m.d_packed, &m.d_packed_pages, &bits);
    if (!stp.ok()) return stp;
    if (bits) {
      m.packed_bits = bits;
      hipFree(m.d_values);  // the 8-byte words are not needed any more
      m.d_values = nullptr;
    }
  }
  t->materialized[c.name] = std::move(m);
  return Status();
}

// ---------------------------------------------------------------------------
// nested (Dremel) scans: CSTableScan::fetchNext with NO_AGGREGATION
// (sql/CSTableScan.cc:187-541) as data-parallel passes.
//
// One output row per slot of the deepest referenced column (the "leaf").  A
// column X at a shallower repetition depth repeats its current value, i.e. row j
// reads X's slot  #{ i <= j : r_leaf[i] <= rlevel_max(X) } - 1.  Undefined slots
// (d < dlevel_max) read as value 0 with tag 0 (an all-zero `SValue()`,
// svalue.cc:154-160 -- NOT a NULL tag).  The number of real slots is where record
// number `num_rows` would start (bit-packed streams are zero-padded).
// ---------------------------------------------------------------------------
// Row-addressable view of one flat column for the AOT kernels (lsm.cc): direct
// page access where the encoding allows it, the cached SoA decode otherwise.
// For STRING columns *strpos receives the device array of (len << 40 | position).
Status table_rt_column(evql_table* t, const std::string& name, RtColumn* out,
                       const uint64_t** strpos) {
  int li = -1;
  for (size_t k = 0; k < t->layout.columns.size(); ++k) {
    if (t->layout.columns[k].name == name) li = int(k);
  }
  if (li < 0) return Status::error(EVQL_EARG, "column not found: " + name);
  const ColumnLayout& cl = t->layout.columns[li];
  if (cl.rlevel_max > 0) return Status::error(EVQL_ENOTSUP, "repeated column: " + name);
  ColAccess c;
  c.name = name;
  c.layout_index = li;
  c.stype = EVQL_T_UINT64;
  *out = RtColumn{};
  out->pages = t->d_pages[li][0];
  if (cl.logical_type == ColumnType::STRING) {
    c.stype = EVQL_T_STRING;
    c.mode = ColAccess::SOA;
    c.string_hash = c.string_bytes = c.has_tags = true;
  } else {
    switch (cl.storage_type) {
      case ColumnEncoding::UINT64_PLAIN:
      case ColumnEncoding::FLOAT_IEEE754: c.mode = ColAccess::PLAIN64; break;
      case ColumnEncoding::UINT32_PLAIN: c.mode = ColAccess::PLAIN32; break;
      case ColumnEncoding::UINT32_BITPACKED:
      case ColumnEncoding::BOOLEAN_BITPACKED: c.mode = ColAccess::BITPACKED; break;
      default: c.mode = ColAccess::SOA;
    }
    if (cl.dlevel_max > 0) {
      c.mode = ColAccess::SOA;
      c.has_tags = true;
    }
  }
  if (c.mode == ColAccess::BITPACKED) {
    uint32_t maxv = 0;
    if (!cl.data_pages.empty()) {
      HIP_TRY(hipMemcpy(&maxv, t->d_image + cl.data_pages[0].offset, 4, hipMemcpyDeviceToHost));
    }
    out->bits = cl.data_pages.empty() ? 0 : bitpack_width(maxv);
  } else if (c.mode == ColAccess::SOA) {
    Status st = materialize_column(t, c, nullptr);
    if (!st.ok()) return st;
    const MaterializedColumn& m = t->materialized[name];
    out->soa = m.d_values;
    out->tags = m.d_tags;
    if (strpos) *strpos = m.d_strpos;
    if (m.packed_bits) {
      c.mode = ColAccess::BITPACKED;
      out->bits = m.packed_bits;
      out->pages = m.d_packed_pages;
      out->base = m.d_packed;
    }
  }
  out->mode = uint32_t(c.mode);
  return Status();
}

static uint64_t level_stream_capacity(const std::vector<PageRef>& pages, uint32_t bits) {
  if (bits == 0) return 0;
  uint64_t blocks = 0;
  for (size_t i = 0; i < pages.size(); ++i) {
    uint64_t bytes = pages[i].size - (i == 0 ? 4 : 0);
    blocks += bytes / (16ull * bits);
  }
  return blocks * 128;
}

static Status stream_bits(evql_table* t, const std::vector<PageRef>& pages, uint32_t* bits) {
  uint32_t maxv = 0;
  if (!pages.empty()) {
    HIP_TRY(hipMemcpy(&maxv, t->d_image + pages[0].offset, 4, hipMemcpyDeviceToHost));
  }
  *bits = pages.empty() ? 0 : bitpack_width(maxv);
  return Status();
}

// slot values of one (possibly repeated / optional) column: vals[slot] = defined
// ? data value : 0, for every slot of its level streams (or `nslots` when the
// column has no definition levels)
static Status nested_slot_values(evql_table* t, int li, uint64_t nslots_flat, uint64_t** out_vals,
                                 uint64_t* out_cap) {
  evql_ctx* ctx = t->ctx;
  hipStream_t s = ctx->stream;
  const ColumnLayout& c = t->layout.columns[li];
  // a column without definition levels is required and top-level: one slot per record
  if (c.dlevel_max == 0) nslots_flat = t->layout.num_rows;
  uint64_t cap = nslots_flat;
  DevBuf<uint8_t> d_tags;
  DevBuf<uint64_t> d_tiles;
  uint64_t nvalues = nslots_flat;
  if (c.dlevel_max > 0) {
    uint32_t dbits = 0;
    Status st = stream_bits(t, c.dlevel_pages, &dbits);
    if (!st.ok()) return st;
    cap = dbits ? level_stream_capacity(c.dlevel_pages, dbits) : nslots_flat;
  }
  const uint64_t capp = padded_rows(cap);
  const uint64_t ntiles = (cap + kDecodeTile - 1) / kDecodeTile;
  HIP_TRY(d_tags.alloc(capp));
  HIP_TRY(d_tiles.alloc((ntiles + 2) * 8));
  if (c.dlevel_max > 0) {
    uint32_t dbits = 0;
    stream_bits(t, c.dlevel_pages, &dbits);
    DevBuf<uint8_t> d_lv;
    HIP_TRY(d_lv.alloc(capp));
    HIP_TRY(hipMemsetAsync(d_lv, 0xff, capp, s));
    if (dbits == 0) {
      // every slot has definition level 0
      HIP_TRY(hipMemsetAsync(d_lv, 0, capp, s));
    } else {
      LevelDecodeArgs la{};
      la.image = t->d_image;
      la.pages = t->d_pages[li][2];
      la.bits = dbits;
      la.nslots = cap;
      la.levels = d_lv;
      for (int k = 0; k < 4; ++k) la.thr[k] = 255;
      HIP_TRY(launch_level_decode(la, s));
    }
    HIP_TRY(launch_defined_from_levels(d_lv, c.dlevel_max, cap, d_tags, d_tiles, s));
    uint64_t* d_total = d_tiles.p + ntiles;
    HIP_TRY(launch_exclusive_scan(d_tiles, ntiles, d_total, s));
    HIP_TRY(hipMemcpyAsync(&nvalues, d_total, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  } else {
    HIP_TRY(hipMemsetAsync(d_tags, 0, capp, s));
    std::vector<uint64_t> offs(ntiles + 1);
    for (uint64_t i = 0; i <= ntiles; ++i) offs[i] = i * kDecodeTile;
    HIP_TRY(hipMemcpyAsync(d_tiles, offs.data(), (ntiles + 1) * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  if (nvalues > fixed_width_capacity(c)) {
    return Status::error(EVQL_EIO, "end of column reached: " + c.name);
  }
  RtColumn src{};
  src.pages = t->d_pages[li][0];
  DevBuf<uint64_t> d_dense;
  switch (c.storage_type) {
    case ColumnEncoding::UINT64_PLAIN:
    case ColumnEncoding::FLOAT_IEEE754:
      src.mode = ColAccess::PLAIN64;
      break;
    case ColumnEncoding::UINT32_PLAIN:
      src.mode = ColAccess::PLAIN32;
      break;
    case ColumnEncoding::UINT32_BITPACKED:
    case ColumnEncoding::BOOLEAN_BITPACKED: {
      src.mode = ColAccess::BITPACKED;
      Status st = stream_bits(t, c.data_pages, &src.bits);
      if (!st.ok()) return st;
      break;
    }
    case ColumnEncoding::UINT64_LEB128: {
      const uint64_t nbytes = uint64_t(c.data_pages.size()) * kPlainPageSize;
      const uint64_t nchunks = (nbytes + kLebChunk - 1) / kLebChunk;
      DevBuf<uint64_t> d_chunks;
      HIP_TRY(d_chunks.alloc((nchunks + 1) * 8));
      HIP_TRY(d_dense.alloc(nvalues * 8));
      if (nchunks) {
        HIP_TRY(launch_leb128_count(t->d_image, t->d_pages[li][0], nbytes, d_chunks, s));
        HIP_TRY(launch_exclusive_scan(d_chunks, nchunks, nullptr, s));
        HIP_TRY(launch_leb128_decode(t->d_image, t->d_pages[li][0], nbytes, d_chunks, nvalues,
                                     d_dense, s));
      }
      HIP_TRY(hipStreamSynchronize(s));
      src.mode = ColAccess::SOA;
      src.soa = d_dense;
      break;
    }
    case ColumnEncoding::STRING_PLAIN: {
      // a string slot's "value" is where its bytes are: (len << 40) | position
      HIP_TRY(d_dense.alloc(std::max<uint64_t>(nvalues, 1) * 8));
      Status st = locate_string_values(t, c, li, nvalues, d_dense);
      if (!st.ok()) return st;
      src.mode = ColAccess::SOA;
      src.soa = d_dense;
      break;
    }
    default:
      return Status::error(EVQL_ENOTSUP, "unsupported nested column encoding");
  }
  DevBuf<uint64_t> d_vals;
  HIP_TRY(d_vals.alloc(capp * 8));
  HIP_TRY(hipMemsetAsync(d_vals, 0, capp * 8, s));
  HIP_TRY(launch_expand_nullable(t->d_image, src, d_tags, d_tiles, cap, d_vals, s));
  HIP_TRY(hipStreamSynchronize(s));
  *out_vals = d_vals.release();
  *out_cap = cap;
  return Status();
}

using LeafLevels = evql_table::LeafLevels;

// number of (r, d, value) slots a repeated column holds for the table's records:
// where record number `num_rows` would start in its (zero-padded) repetition levels
static Status exact_slot_count(evql_table* t, int li, uint64_t* out) {
  const ColumnLayout& c = t->layout.columns[li];
  const uint64_t nrec = t->layout.num_rows;
  if (c.rlevel_max == 0) {
    *out = nrec;
    return Status();
  }
  hipStream_t s = t->ctx->stream;
  uint32_t rbits = 0;
  Status st = stream_bits(t, c.rlevel_pages, &rbits);
  if (!st.ok()) return st;
  if (rbits == 0) return Status::error(EVQL_ENOTSUP, "repeated column without repetition levels");
  const uint64_t cap = level_stream_capacity(c.rlevel_pages, rbits);
  const uint64_t capp = padded_rows(cap);
  const uint64_t ntiles = (cap + kDecodeTile - 1) / kDecodeTile;
  DevBuf<uint8_t> d_lv;
  DevBuf<uint64_t> d_cnt, d_n;
  HIP_TRY(d_lv.alloc(capp));
  HIP_TRY(hipMemsetAsync(d_lv, 0xff, capp, s));
  HIP_TRY(d_cnt.alloc((ntiles + 2) * 8));
  HIP_TRY(d_n.alloc(8));
  LevelDecodeArgs la{};
  la.image = t->d_image;
  la.pages = t->d_pages[li][1];
  la.bits = rbits;
  la.nslots = cap;
  la.levels = d_lv;
  for (int k = 0; k < 4; ++k) la.thr[k] = 255;
  la.counts[0] = d_cnt;
  la.thr[0] = 0;
  HIP_TRY(launch_level_decode(la, s));
  HIP_TRY(launch_exclusive_scan(d_cnt, ntiles, nullptr, s));
  uint64_t n = cap;
  HIP_TRY(hipMemcpyAsync(d_n, &n, 8, hipMemcpyHostToDevice, s));
  HIP_TRY(launch_find_nth(d_lv, d_cnt, cap, 0, nrec, d_n, s));
  HIP_TRY(hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  *out = n;
  return Status();
}

// flattens `cols` (all of one ancestor chain) to one value per leaf slot:
// (*flat)[i] is borrowed from the table's nested cache
static Status materialize_nested(evql_query* q, const std::vector<ColAccess>& cols,
                                 std::vector<uint64_t*>* flat_out, uint64_t* nrows_out,
                                 LeafLevels* keep, std::vector<uint64_t*>* strpos_out = nullptr) {
  evql_table* t = q->table;
  evql_ctx* ctx = q->ctx;
  hipStream_t s = ctx->stream;
  struct {
    const std::vector<ColAccess>& cols;
  } kp{cols};
  std::vector<uint64_t*>& nested_flat = *flat_out;
  const uint64_t nrec = t->layout.num_rows;
  nested_flat.assign(kp.cols.size(), nullptr);
  if (strpos_out) strpos_out->assign(kp.cols.size(), nullptr);
  // a string column's row value is (len << 40 | position); the kernels group and
  // compare on its hash (`flat`) and read the bytes through the position (`strpos`)
  auto publish = [&](size_t i, const evql_table::NestedFlat& e) {
    if (kp.cols[i].string_hash) {
      nested_flat[i] = e.d_hash;
      if (strpos_out) (*strpos_out)[i] = e.d_values;
    } else {
      nested_flat[i] = e.d_values;
    }
  };
  if (kp.cols.empty()) {
    *nrows_out = nrec;  // fetchNextWithoutColumns: one row per record
    return Status();
  }
  // leaf = deepest referenced column
  int leaf = 0;
  for (size_t i = 0; i < kp.cols.size(); ++i) {
    if (t->layout.columns[kp.cols[i].layout_index].rlevel_max >
        t->layout.columns[kp.cols[leaf].layout_index].rlevel_max) {
      leaf = int(i);
    }
  }
  const ColumnLayout& lc = t->layout.columns[kp.cols[leaf].layout_index];
  const int leaf_li = kp.cols[leaf].layout_index;
  q->nested_leaf = leaf_li;
  {
    // every column already flattened for this leaf by an earlier operator?
    bool all = keep == nullptr || lc.rlevel_max == 0 || t->leaf_cache.count(leaf_li);
    for (const auto& c : kp.cols) all = all && t->nested_cache.count({c.layout_index, leaf_li});
    if (all) {
      if (keep && lc.rlevel_max > 0) *keep = t->leaf_cache[leaf_li];
      for (size_t i = 0; i < kp.cols.size(); ++i) {
        const auto& e = t->nested_cache[{kp.cols[i].layout_index, leaf_li}];
        publish(i, e);
        *nrows_out = e.nflat;
      }
      return Status();
    }
  }
  uint64_t nflat = nrec;
  uint64_t leaf_cap = 0;
  DevBuf<uint8_t> d_leaf_levels;
  std::vector<uint32_t> thr_levels;       // distinct parent rlevel_max values
  struct OwnedList {                       // scanned per-tile counts per threshold
    std::vector<uint64_t*> v;
    ~OwnedList() {
      for (auto* p : v) hipFree(p);
    }
  } thr_offsets_owner;
  std::vector<uint64_t*>& thr_offsets = thr_offsets_owner.v;
  if (lc.rlevel_max > 0) {
    uint32_t rbits = 0;
    Status st = stream_bits(t, lc.rlevel_pages, &rbits);
    if (!st.ok()) return st;
    const uint64_t cap = level_stream_capacity(lc.rlevel_pages, rbits);
    leaf_cap = cap;
    const uint64_t capp = padded_rows(cap);
    const uint64_t ntiles = (cap + kDecodeTile - 1) / kDecodeTile;
    thr_levels.push_back(0);
    for (const auto& c : kp.cols) {
      uint32_t rm = t->layout.columns[c.layout_index].rlevel_max;
      if (rm >= lc.rlevel_max) continue;
      bool seen = false;
      for (auto x : thr_levels) seen = seen || x == rm;
      if (!seen) thr_levels.push_back(rm);
    }
    if (thr_levels.size() > 4) {
      return Status::error(EVQL_ENOTSUP, "more than four repetition depths in one nested scan");
    }
    HIP_TRY(d_leaf_levels.alloc(capp));
    HIP_TRY(hipMemsetAsync(d_leaf_levels, 0xff, capp, s));
    LevelDecodeArgs la{};
    la.image = t->d_image;
    la.pages = t->d_pages[kp.cols[leaf].layout_index][1];
    la.bits = rbits;
    la.nslots = cap;
    la.levels = d_leaf_levels;
    for (int k = 0; k < 4; ++k) la.thr[k] = 255;
    for (size_t k = 0; k < thr_levels.size(); ++k) {
      uint64_t* d = nullptr;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), (ntiles + 2) * 8));
      thr_offsets.push_back(d);
      la.counts[k] = d;
      la.thr[k] = thr_levels[k];
    }
    if (rbits == 0) {
      return Status::error(EVQL_ENOTSUP, "repeated column without repetition levels");
    }
    HIP_TRY(launch_level_decode(la, s));
    for (size_t k = 0; k < thr_levels.size(); ++k) {
      HIP_TRY(launch_exclusive_scan(thr_offsets[k], ntiles, nullptr, s));
    }
    // number of real slots = start of record number `nrec`
    DevBuf<uint64_t> d_n;
    HIP_TRY(d_n.alloc(8));
    HIP_TRY(hipMemcpyAsync(d_n, &cap, 8, hipMemcpyHostToDevice, s));
    HIP_TRY(launch_find_nth(d_leaf_levels, thr_offsets[0], cap, 0, nrec, d_n, s));
    HIP_TRY(hipMemcpyAsync(&nflat, d_n, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  *nrows_out = nflat;
  const uint8_t* leaf_levels = d_leaf_levels.p;
  const uint64_t flatp = padded_rows(nflat);
  for (size_t i = 0; i < kp.cols.size(); ++i) {
    // the same column referenced twice shares one buffer
    const int li = kp.cols[i].layout_index;
    {
      // (the same column referenced twice, or flattened by an earlier operator)
      auto hit = t->nested_cache.find({li, leaf_li});
      if (hit != t->nested_cache.end()) {
        publish(i, hit->second);
        continue;
      }
    }
    const ColumnLayout& c = t->layout.columns[li];
    DevBuf<uint64_t> d_vals;
    uint64_t cap = 0;
    Status st = nested_slot_values(t, li, nflat, &d_vals.p, &cap);
    if (!st.ok()) return st;
    if (c.rlevel_max > 0 && li != leaf_li) {
      // exact ancestor-chain check (the planner's is on names only): a column on the
      // leaf's chain has one slot per leaf slot whose repetition level does not
      // exceed the column's depth.  A sibling repeated group passes only by
      // coincidence of every count.
      uint64_t own = 0;
      st = exact_slot_count(t, li, &own);
      if (!st.ok()) return st;
      bool chain = false;
      if (c.rlevel_max >= lc.rlevel_max) {
        chain = own == nflat;
      } else {
        size_t k = 0;
        while (thr_levels[k] != c.rlevel_max) ++k;
        // the own-th (0-based) leaf slot with r <= depth must be the first padding slot
        DevBuf<uint64_t> d_n;
        HIP_TRY(d_n.alloc(8));
        uint64_t at = ~0ull;
        HIP_TRY(hipMemcpyAsync(d_n, &at, 8, hipMemcpyHostToDevice, s));
        HIP_TRY(launch_find_nth(leaf_levels, thr_offsets[k], leaf_cap, c.rlevel_max, own, d_n, s));
        HIP_TRY(hipMemcpyAsync(&at, d_n, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        // (a leaf stream without a padding slot cannot be probed this way)
        chain = at == nflat || nflat == leaf_cap;
      }
      if (!chain) {
        return Status::error(EVQL_ENOTSUP, "nested columns from different repeated groups");
      }
    }
    if (c.rlevel_max >= lc.rlevel_max) {
      if (padded_rows(cap) < flatp) {
        // level streams shorter than the leaf's: not the same ancestor chain
        return Status::error(EVQL_ENOTSUP, "nested columns from different repeated groups");
      }
      t->nested_cache[{li, leaf_li}] = evql_table::NestedFlat{d_vals.release(), nflat, nullptr};
    } else {
      size_t k = 0;
      while (thr_levels[k] != c.rlevel_max) ++k;
      DevBuf<uint64_t> d_flat;
      HIP_TRY(d_flat.alloc(flatp * 8));
      HIP_TRY(hipMemsetAsync(d_flat, 0, flatp * 8, s));
      HIP_TRY(launch_flatten_parent(leaf_levels, thr_offsets[k], c.rlevel_max, nflat, d_vals,
                                    d_flat, s));
      HIP_TRY(hipStreamSynchronize(s));
      t->nested_cache[{li, leaf_li}] = evql_table::NestedFlat{d_flat.release(), nflat, nullptr};
    }
    evql_table::NestedFlat& e = t->nested_cache[{li, leaf_li}];
    if (kp.cols[i].string_hash) {
      // undefined slots carry strpos 0: the empty string (an all-zero SValue read as
      // a STRING, CSTableScan.cc:224-246)
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e.d_hash), flatp * 8));
      HIP_TRY(hipMemsetAsync(e.d_hash, 0, flatp * 8, s));
      HIP_TRY(launch_string_hash(t->d_image, t->d_pages[li][0], e.d_values, nflat, e.d_hash, s));
      HIP_TRY(hipStreamSynchronize(s));
    }
    publish(i, e);
  }
  if (keep && lc.rlevel_max > 0) {
    auto hit = t->leaf_cache.find(leaf_li);
    if (hit == t->leaf_cache.end()) {
      LeafLevels ll;
      ll.levels = d_leaf_levels.release();
      ll.rec_offsets = thr_offsets[0];  // threshold 0 comes first
      thr_offsets[0] = nullptr;
      hit = t->leaf_cache.emplace(leaf_li, ll).first;
    }
    *keep = hit->second;
  }
  return Status();
}

// Columns of SIBLING repeated groups (or of groups at different depths that share no
// chain): CSTableScan::fetchNext zips them level by level (CSTableScan.cc:187-541) -- a row
// per step of its column automaton; a group that has run out of slots reads the all-zero
// SValue from then on (:511-515), columns above the fetch level keep their value.  One
// thread replays that automaton per record (k_zip_rows), first counting the record's rows,
// then writing the slot every (column, row) reads; the flattened columns are gathered from
// the per-slot values.  Owned by the operator (not cached on the table).
static Status materialize_nested_zip(evql_query* q, const std::vector<ColAccess>& cols,
                                     std::vector<uint64_t*>* flat_out, uint64_t* nrows_out,
                                     std::vector<uint64_t*>* strpos_out) {
  evql_table* t = q->table;
  hipStream_t s = q->ctx->stream;
  const uint64_t nrec = t->layout.num_rows;
  const size_t nc = cols.size();
  if (nc > kMaxZipCols) return Status::error(EVQL_ENOTSUP, "too many columns in a zipped nested scan");
  flat_out->assign(nc, nullptr);
  if (strpos_out) strpos_out->assign(nc, nullptr);
  q->nested_leaf = -1;
  if (nrec == 0) {
    *nrows_out = 0;
    return Status();
  }
  ZipArgs za{};
  za.nrec = nrec;
  za.ncols = uint32_t(nc);
  std::vector<DevBuf<uint64_t>> d_vals(nc), d_starts(nc), d_idx(nc);
  std::vector<DevBuf<uint8_t>> d_levels(nc);
  std::map<int, size_t> first_use;  // layout index -> first scan column that decoded it
  for (size_t i = 0; i < nc; ++i) {
    const int li = cols[i].layout_index;
    const ColumnLayout& c = t->layout.columns[li];
    za.rmax[i] = c.rlevel_max;
    auto seen = first_use.find(li);
    if (seen != first_use.end()) {
      const size_t j = seen->second;
      za.levels[i] = za.levels[j];
      za.starts[i] = za.starts[j];
      continue;
    }
    first_use[li] = i;
    uint64_t cap = 0;
    Status st = nested_slot_values(t, li, nrec, &d_vals[i].p, &cap);
    if (!st.ok()) return st;
    if (c.rlevel_max == 0) continue;  // one slot per record: levels / starts stay NULL
    uint32_t rbits = 0;
    st = stream_bits(t, c.rlevel_pages, &rbits);
    if (!st.ok()) return st;
    if (rbits == 0) return Status::error(EVQL_ENOTSUP, "repeated column without repetition levels");
    const uint64_t lcap = level_stream_capacity(c.rlevel_pages, rbits);
    const uint64_t lcapp = padded_rows(lcap);
    const uint64_t ntiles = (lcap + kDecodeTile - 1) / kDecodeTile;
    DevBuf<uint64_t> d_cnt;
    HIP_TRY(d_levels[i].alloc(lcapp));
    HIP_TRY(hipMemsetAsync(d_levels[i], 0xff, lcapp, s));
    HIP_TRY(d_cnt.alloc((ntiles + 2) * 8));
    LevelDecodeArgs la{};
    la.image = t->d_image;
    la.pages = t->d_pages[li][1];
    la.bits = rbits;
    la.nslots = lcap;
    la.levels = d_levels[i];
    for (int k = 0; k < 4; ++k) la.thr[k] = 255;
    la.counts[0] = d_cnt;
    la.thr[0] = 0;
    HIP_TRY(launch_level_decode(la, s));
    HIP_TRY(launch_exclusive_scan(d_cnt, ntiles, nullptr, s));
    HIP_TRY(d_starts[i].alloc((nrec + 2) * 8));
    // (a level stream without zero padding: record `nrec` would start at its end)
    HIP_TRY(hipMemcpyAsync(d_starts[i].p + nrec, &lcap, 8, hipMemcpyHostToDevice, s));
    HIP_TRY(launch_record_starts(d_levels[i], d_cnt, lcap, d_starts[i], nrec + 1, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (cap < lcap && c.dlevel_max > 0) {
      return Status::error(EVQL_EIO, "level streams of different length: " + c.name);
    }
    za.levels[i] = d_levels[i];
    za.starts[i] = d_starts[i];
  }
  DevBuf<uint64_t> d_rows;
  DevBuf<ZipArgs> d_args;
  HIP_TRY(d_rows.alloc((nrec + 2) * 8));
  HIP_TRY(d_args.alloc(sizeof(ZipArgs)));
  za.rows = d_rows;
  HIP_TRY(hipMemcpyAsync(d_args, &za, sizeof(ZipArgs), hipMemcpyHostToDevice, s));
  HIP_TRY(launch_zip_rows(d_args, nrec, 0, s));
  uint64_t nflat = 0;
  HIP_TRY(launch_exclusive_scan(d_rows, nrec, d_rows.p + nrec, s));
  HIP_TRY(hipMemcpyAsync(&nflat, d_rows.p + nrec, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  *nrows_out = nflat;
  const uint64_t flatp = padded_rows(nflat);
  for (size_t i = 0; i < nc; ++i) {
    HIP_TRY(d_idx[i].alloc(std::max<uint64_t>(nflat, 1) * 8));
    za.idx[i] = d_idx[i];
  }
  HIP_TRY(hipMemcpyAsync(d_args, &za, sizeof(ZipArgs), hipMemcpyHostToDevice, s));
  HIP_TRY(launch_zip_rows(d_args, nrec, 1, s));
  for (size_t i = 0; i < nc; ++i) {
    const int li = cols[i].layout_index;
    const size_t src = first_use[li];
    DevBuf<uint64_t> d_flat;
    HIP_TRY(d_flat.alloc(flatp * 8));
    HIP_TRY(hipMemsetAsync(d_flat, 0, flatp * 8, s));
    HIP_TRY(launch_zip_gather(d_vals[src], d_idx[i], nflat, d_flat, s));
    if (cols[i].string_hash) {
      // the flattened words are (len << 40 | position); a reset slot reads strpos 0: ""
      DevBuf<uint64_t> d_hash;
      HIP_TRY(d_hash.alloc(flatp * 8));
      HIP_TRY(hipMemsetAsync(d_hash, 0, flatp * 8, s));
      HIP_TRY(launch_string_hash(t->d_image, t->d_pages[li][0], d_flat, nflat, d_hash, s));
      (*flat_out)[i] = d_hash;
      if (strpos_out) (*strpos_out)[i] = d_flat;
      q->nested_owned.push_back(d_hash.release());
      q->nested_owned.push_back(d_flat.release());
    } else {
      (*flat_out)[i] = d_flat;
      q->nested_owned.push_back(d_flat.release());
    }
  }
  HIP_TRY(hipStreamSynchronize(s));
  return Status();
}

// CSTableScan with AGGREGATE_WITHIN_RECORD_FLAT (CSTableScan.cc:440-487): one
// output row per record holding the scan select list's aggregates over the
// record's flattened rows.  select_list_[i] accumulates on the rows whose fetch
// level is <= its rep_level (:442); with every column on one ancestor chain the
// fetch level of a row is the leaf's repetition level of that slot.
static Status materialize_within_record(evql_query* q) {
  evql_table* t = q->table;
  hipStream_t s = q->ctx->stream;
  const uint64_t nrec = t->layout.num_rows;
  if (q->wr_aggs.size() > kMaxWithinAggs) {
    return Status::error(EVQL_ENOTSUP, "too many WITHIN RECORD aggregates");
  }
  std::vector<uint64_t*> flat;
  uint64_t nflat = 0;
  LeafLevels leaf;
  Status st = materialize_nested(q, q->wr_cols, &flat, &nflat, &leaf);
  if (!st.ok()) return st;
  WithinRecordArgs a{};
  a.leaf_levels = leaf.levels;
  a.rec_offsets = leaf.rec_offsets;
  a.nflat = nflat;
  a.nrec = nrec;
  a.n = uint32_t(q->wr_aggs.size());
  const uint64_t recp = padded_rows(nrec);
  q->nested_flat.assign(q->wr_aggs.size(), nullptr);
  for (size_t e = 0; e < q->wr_aggs.size(); ++e) {
    const evql_query::WithinAgg& w = q->wr_aggs[e];
    DevBuf<uint64_t> d_out;
    HIP_TRY(d_out.alloc(recp * 8));
    // (every record is stored by the kernel: only the padding behind them is cleared)
    if (recp > nrec) HIP_TRY(hipMemsetAsync(d_out.p + nrec, 0, (recp - nrec) * 8, s));
    if (leaf.levels) {
      const uint64_t ntiles = (nflat + kDecodeTile - 1) / kDecodeTile;
      DevBuf<uint64_t> d_head;
      HIP_TRY(d_head.alloc(std::max<uint64_t>(ntiles, 1) * 8));
      a.tile_head[e] = d_head;
      q->nested_owned.push_back(d_head.release());
    }
    a.src[e] = w.col >= 0 ? flat[w.col] : nullptr;
    a.lit[e] = w.lit;
    a.level[e] = w.level;
    a.is_count[e] = w.is_count ? 1 : 0;
    a.out[e] = d_out;
    q->nested_flat[e] = d_out;
    q->nested_owned.push_back(d_out.release());
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipEventRecord(e0, s));
  HIP_TRY(launch_within_record(a, s));
  HIP_TRY(hipEventRecord(e1, s));
  HIP_TRY(hipStreamSynchronize(s));
  float wms = 0;
  hipEventElapsedTime(&wms, e0, e1);
  q->within_record_ms = wms;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  q->nested_rows = nrec;
  return Status();
}

// ---------------------------------------------------------------------------
// query execution
// ---------------------------------------------------------------------------
static uint64_t word_identity(int op) {
  switch (op) {
    case 2: return 0xFFFFFFFFFFFFFFFFull;
    case 3: return 0ull;
    case 4: return 0x7FFFFFFFFFFFFFFFull;
    case 5: return 0x8000000000000000ull;
    case 6: return 0x7FF0000000000000ull;
    case 7: return 0xFFF0000000000000ull;
    default: return 0ull;
  }
}

// ---------------------------------------------------------------------------
// EVQL_FLOAT_SUM_EXACT: bound of |argument| and the quantum of every exact sum
// ---------------------------------------------------------------------------
// upper bound of |e| given upper bounds of |column i|; +inf = unknown
static double expr_abs_bound(const ExprPtr& e, const std::vector<double>& colmax) {
  const double inf = std::numeric_limits<double>::infinity();
  switch (e->kind) {
    case Expr::INPUT:
      return e->input < colmax.size() ? colmax[e->input] : inf;
    case Expr::LITERAL:
      switch (e->type) {
        case EVQL_T_UINT64: case EVQL_T_TIMESTAMP64: return double(e->lit_bits);
        case EVQL_T_INT64: return std::fabs(double(int64_t(e->lit_bits)));
        case EVQL_T_FLOAT64: {
          double d;
          memcpy(&d, &e->lit_bits, 8);
          return std::fabs(d);
        }
        case EVQL_T_BOOL: return 1.0;
        default: return inf;
      }
    case Expr::IF:
      return std::max(expr_abs_bound(e->args[1], colmax), expr_abs_bound(e->args[2], colmax));
    case Expr::CALL: {
      std::vector<double> b;
      for (const auto& a : e->args) b.push_back(expr_abs_bound(a, colmax));
      switch (e->family) {
        case EVQL_FAM_ADD: case EVQL_FAM_SUB: return b[0] + b[1];
        case EVQL_FAM_MUL: return b[0] * b[1];
        case EVQL_FAM_MOD: return b[0];
        case EVQL_FAM_DIV:
          if (e->type != EVQL_T_FLOAT64) return b[0];
          if (e->args[1]->kind == Expr::LITERAL) {
            double d;
            memcpy(&d, &e->args[1]->lit_bits, 8);
            if (d != 0.0) return b[0] / std::fabs(d);
          }
          return inf;
        case EVQL_FAM_TO_INT64: case EVQL_FAM_TO_TIMESTAMP64: return b[0];
        case EVQL_FAM_CMP: case EVQL_FAM_EQ: case EVQL_FAM_NEQ: case EVQL_FAM_LT:
        case EVQL_FAM_LTE: case EVQL_FAM_GT: case EVQL_FAM_GTE: case EVQL_FAM_LOGICAL_AND:
        case EVQL_FAM_LOGICAL_OR: case EVQL_FAM_NEG:
          return 1.0;
        default: return inf;
      }
    }
    default:
      return inf;
  }
}

// maximum |value| of scan column i as the kernel sees it (cached per table column)
static Status column_abs_max(evql_query* q, size_t i, double* out) {
  evql_table* t = q->table;
  hipStream_t s = q->ctx->stream;
  const ColAccess& c = q->kp.cols[i];
  if (c.string_hash) {
    *out = std::numeric_limits<double>::infinity();
    return Status();
  }
  if (c.dict_code) {  // dense codes 0 .. n_codes - 1
    *out = double(t->dicts[c.name].n_codes - 1);
    return Status();
  }
  const bool is_float = c.stype == EVQL_T_FLOAT64 && !c.from_uint_to_float;
  const std::string key = c.name + (is_float ? "#f" : "#u");
  if (!q->nested) {
    auto hit = t->col_absmax.find(key);
    if (hit != t->col_absmax.end()) {
      *out = hit->second;
      return Status();
    }
  }
  RtColumn rc{};
  rc.pages = c.layout_index >= 0 ? t->d_pages[c.layout_index][0] : nullptr;
  rc.mode = c.mode;
  rc.bits = c.bits;
  uint64_t n = t->layout.num_rows;
  if (q->nested) {
    rc.mode = ColAccess::SOA;  // (the 8-byte words stay beside a packed copy)
    rc.soa = q->nested_flat[i];
    n = q->nested_rows;
  } else if (c.packed) {
    const MaterializedColumn& m = t->materialized[c.name];
    rc.pages = m.d_packed_pages;
    rc.base = m.d_packed;
  } else if (c.mode == ColAccess::SOA) {
    rc.soa = t->materialized[c.name].d_values;
  }
  DevBuf<uint64_t> d_max;
  HIP_TRY(d_max.alloc(8));
  HIP_TRY(hipMemsetAsync(d_max, 0, 8, s));
  HIP_TRY(launch_column_abs_max(t->d_image, rc, n, is_float ? 1 : 0, d_max, s));
  uint64_t bits = 0;
  HIP_TRY(hipMemcpyAsync(&bits, d_max, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  double m;
  if (is_float) {
    memcpy(&m, &bits, 8);  // (NaN / inf order above every finite |x| as integers)
    if (!(m == m)) m = std::numeric_limits<double>::infinity();
  } else {
    m = double(bits);
  }
  if (!q->nested) t->col_absmax[key] = m;
  *out = m;
  return Status();
}

static Status choose_exact_sum_scales(evql_query* q) {
  std::vector<double> colmax(q->kp.cols.size(), std::numeric_limits<double>::infinity());
  bool have = false;
  for (const auto& a : q->kp.aggs) {
    if (a.exact_index < 0) continue;
    double bound = q->float_sum_bound;
    if (!(bound > 0)) {
      if (!have) {
        for (size_t i = 0; i < colmax.size(); ++i) {
          Status st = column_abs_max(q, i, &colmax[i]);
          if (!st.ok()) return st;
        }
        have = true;
      }
      bound = a.arg ? expr_abs_bound(a.arg, colmax) : 0.0;
      if (!std::isfinite(bound)) {
        return Status::error(EVQL_ENOTSUP, "exact float sum: no finite bound of the argument "
                                           "follows from the table; pass float_sum_bound");
      }
    }
    // quantum 2^e with bound * 2^-e < 2^61: |q| < 2^61, the high parts (|q| >> 31
    // < 2^30) and the low parts (< 2^31) of up to 2^32 rows add up inside 64 bits
    int ex = 0;
    std::frexp(bound > 0 ? bound : 1.0, &ex);  // bound < 2^ex
    q->fsum_exp[a.exact_index] = ex - 61;
    q->fsum_bound[a.exact_index] = bound;
  }
  return Status();
}

// ---------------------------------------------------------------------------
// partitioned path: which tuple members fit 32 bits
// ---------------------------------------------------------------------------
// upper bound of the unsigned value of e given upper bounds of the columns; +inf where
// the value may wrap or is not an unsigned integer
static double expr_unsigned_bound(const ExprPtr& e, const std::vector<double>& colmax) {
  const double inf = std::numeric_limits<double>::infinity();
  const bool uns = e->type == EVQL_T_UINT64 || e->type == EVQL_T_TIMESTAMP64 || e->type == EVQL_T_BOOL;
  if (!uns) return inf;
  switch (e->kind) {
    case Expr::INPUT:
      return e->input < colmax.size() ? colmax[e->input] : inf;
    case Expr::LITERAL:
      return e->type == EVQL_T_BOOL ? 1.0 : double(e->lit_bits);
    case Expr::IF:
      return std::max(expr_unsigned_bound(e->args[1], colmax), expr_unsigned_bound(e->args[2], colmax));
    case Expr::CALL: {
      if (e->type == EVQL_T_BOOL) return 1.0;
      std::vector<double> b;
      for (const auto& a : e->args) b.push_back(expr_unsigned_bound(a, colmax));
      switch (e->family) {
        case EVQL_FAM_ADD: return b[0] + b[1];  // (< 2^53: exact in a double; larger sums
        case EVQL_FAM_MUL: return b[0] * b[1];  //  are far beyond the 2^32 threshold)
        case EVQL_FAM_MOD: case EVQL_FAM_DIV: return b[0];
        default: return inf;
      }
    }
    default:
      return inf;
  }
}

// Sets narrow_ident / narrow_first_row / AggPlan::narrow_arg from the maxima of the
// referenced columns (one streaming pass per table column, cached): a member that
// provably stays below 2^32 - 1 travels as 4 bytes through scatter / refine / aggregate.
static Status choose_tuple_widths(evql_query* q) {
  KernelPlan& kp = q->kp;
  const double lim = 4294967295.0;  // strictly below: 32 ones are a minimum's identity
  std::vector<double> colmax(kp.cols.size(), std::numeric_limits<double>::infinity());
  for (size_t i = 0; i < colmax.size(); ++i) {
    const ColAccess& c = kp.cols[i];
    if (c.string_hash || (c.stype == EVQL_T_FLOAT64 && !c.from_uint_to_float)) continue;
    if (c.stype != EVQL_T_UINT64 && c.stype != EVQL_T_TIMESTAMP64 && c.stype != EVQL_T_BOOL) continue;
    Status st = column_abs_max(q, i, &colmax[i]);
    if (!st.ok()) return st;
  }
  kp.narrow_ident = kp.key_mode == KEY_EXACT && expr_unsigned_bound(kp.group[0], colmax) < lim;
  const uint64_t nrows = q->nested ? q->nested_rows : q->table->layout.num_rows;
  kp.narrow_first_row = nrows < (1ull << 32);
  for (auto& a : kp.aggs) {
    a.narrow_arg = a.arg && expr_unsigned_bound(a.arg, colmax) < lim;
  }
  return Status();
}

static Status compile_plan_kernels(evql_query* q);

static Status apply_where_resets(evql_query* q, const evql_table::LeafLevels& leaf);

Status query_prepare(evql_query* q) {
  evql_table* t = q->table;
  evql_table::LeafLevels where_leaf;
  if (q->within_record) {
    Status st = materialize_within_record(q);
    if (!st.ok()) return st;
  } else if (q->nested) {
    Status st;
    if (!q->nested_siblings) {
      st = materialize_nested(q, q->kp.cols, &q->nested_flat, &q->nested_rows,
                              q->nested_where_mixed ? &where_leaf : nullptr, &q->nested_strpos);
      // (the planner's chain check reads names only: groups that merely look like one
      // chain are caught by the slot counts)
      if (!st.ok() && st.code == EVQL_ENOTSUP && !q->nested_where_mixed &&
          st.msg.find("different repeated groups") != std::string::npos) {
        q->nested_siblings = true;
      } else if (!st.ok()) {
        return st;
      }
    }
    if (q->nested_siblings) {
      st = materialize_nested_zip(q, q->kp.cols, &q->nested_flat, &q->nested_rows, &q->nested_strpos);
      if (!st.ok()) return st;
    }
  }
  // resolve bit widths and materialise SoA columns
  bool repacked = false;
  q->nested_packed.assign(q->kp.cols.size(), evql_query::PackedSource{});
  if (q->nested && !q->within_record && !q->nested_where_mixed && q->nested_leaf >= 0) {
    // The fused kernel streams the flattened columns; like required LEB128 columns they
    // are kept once more as bit-packed pages of 8 / 16 / 32 bits where their maximum
    // fits (config 5: 1 + 4 bytes per row instead of 8 + 8).  Not for string hashes,
    // nor when WHERE resets rewrite the columns per query (apply_where_resets).
    for (size_t i = 0; i < q->kp.cols.size(); ++i) {
      ColAccess& c = q->kp.cols[i];
      if (c.string_hash || c.stype == EVQL_T_FLOAT64) continue;
      auto hit = t->nested_cache.find({c.layout_index, q->nested_leaf});
      if (hit == t->nested_cache.end()) continue;
      evql_table::NestedFlat& e = hit->second;
      if (!e.pack_tried) {
        e.pack_tried = true;
        Status stp = pack_narrow(q->ctx->stream, e.d_values, e.nflat, &e.d_packed, &e.d_packed_pages,
                                 &e.packed_bits);
        if (!stp.ok()) return stp;
      }
      if (!e.packed_bits) continue;
      c.mode = ColAccess::BITPACKED;
      c.bits = e.packed_bits;
      c.packed = true;
      q->nested_packed[i].base = e.d_packed;
      q->nested_packed[i].pages = e.d_packed_pages;
      repacked = true;
    }
  }
  for (auto& c : q->kp.cols) {
    if (q->nested) break;
    const ColumnLayout& cl = t->layout.columns[c.layout_index];
    if (c.mode == ColAccess::BITPACKED) {
      uint32_t maxv = 0;
      if (!cl.data_pages.empty()) {
        HIP_TRY(hipMemcpy(&maxv, t->d_image + cl.data_pages[0].offset, 4, hipMemcpyDeviceToHost));
      }
      c.bits = cl.data_pages.empty() ? 0 : bitpack_width(maxv);
    } else if (c.mode == ColAccess::SOA) {
      Status st = materialize_column(t, c, nullptr);
      if (!st.ok()) return st;
      const MaterializedColumn& m = t->materialized[c.name];
      if (m.packed_bits) {  // LEB128 kept as narrow bit-packed pages
        c.mode = ColAccess::BITPACKED;
        c.bits = m.packed_bits;
        c.packed = true;
        repacked = true;
      }
    }
  }
  if (q->dict_candidate >= 0 && !q->nested) {
    // a STRING key with a usable dictionary: the kernels group by its 32-bit codes.
    // (Here, behind the loop above: the record-level copy of the plan must know the
    // resolved access modes of the other columns -- first-row gathers read them.)
    KernelPlan& kp = q->kp;
    const int ki = q->dict_candidate;
    StringDict* dict = nullptr;
    Status std_ = table_string_dict(t, kp.cols[ki].layout_index, &dict);
    if (!std_.ok()) return std_;
    if (dict->usable) {
      q->rkp = kp;  // what the group records look like outside the scan
      ColAccess code = kp.cols[ki];
      code.stype = EVQL_T_UINT64;
      code.mode = ColAccess::PLAIN32;
      code.has_tags = false;
      code.string_hash = code.string_bytes = false;
      code.dict_code = true;
      kp.cols[ki] = code;
      auto in = std::make_shared<Expr>();
      in->kind = Expr::INPUT;
      in->type = EVQL_T_UINT64;
      in->input = uint32_t(ki);
      kp.group[0] = in;
      kp.key_mode = KEY_EXACT;
      kp.need_first_row = false;
      q->dict_key = true;
      choose_launch_shape(&kp, q->groups_hint);
    }
  }
  if (repacked && !q->kp.partitioned) {
    // the access modes changed: block / unroll / LDS table are chosen again
    choose_launch_shape(&q->kp, q->groups_hint);
  }
  if (q->kp.n_exact > 0) {
    Status stb = choose_exact_sum_scales(q);
    if (!stb.ok()) return stb;
  }
  Status st = compile_plan_kernels(q);
  if (!st.ok()) return st;
  // (two allocations on purpose: with the status words and the counters in one 128-byte
  // line -- tried, to read both back with one copy -- the scan kernel's per-tile poll of
  // status[0] shared its line with the counter atomics: config 3 over 16-bit pages
  // 0.36 -> 0.58 ms)
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_status), 16));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_counters), 64));
  if (q->nested_where_mixed) {
    st = apply_where_resets(q, where_leaf);
    if (!st.ok()) return st;
  }
  HIP_TRY(hipEventCreate(&q->ev0));
  HIP_TRY(hipEventCreate(&q->ev1));
  if (!q->row_filter_host.empty()) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_row_filter), q->row_filter_host.size() + 16));
    HIP_TRY(hipMemcpy(q->d_row_filter, q->row_filter_host.data(), q->row_filter_host.size(),
                      hipMemcpyHostToDevice));
  }
  return Status();
}

static Status alloc_gtab(evql_query* q, uint64_t gcap) {
  if (q->d_gtab) {
    hipFree(q->d_gtab);
    q->d_gtab = nullptr;
  }
  q->gcap = gcap;
  const uint64_t stride = gcap + 8;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_gtab),
                    stride * uint64_t(q->kp.words_per_slot()) * 8));
  return Status();
}

static Status alloc_gtab(evql_query* q, uint64_t gcap);
Status query_launch(evql_query* q);
Status query_finish(evql_query* q);

// (re)compiles the fused kernel(s) of q->kp and sizes the persistent grid
static Status compile_plan_kernels(evql_query* q) {
  evql_ctx* ctx = q->ctx;
  if (q->kp.partitioned) {
    Status stw = choose_tuple_widths(q);
    if (!stw.ok()) return stw;
    // two partition levels: without the count pass, unless a coarse bucket overflowed its
    // slack before (skewed keys) or the tuple buffers have to be sized by an exact count
    // (very large scans with a selective predicate, query_launch)
    evql_table* t = q->table;
    const uint64_t nrows = q->nested ? q->nested_rows : t->layout.num_rows;
    const uint64_t begin = std::min(q->row_begin, nrows);
    const uint64_t end = q->row_end ? std::min(q->row_end, nrows) : nrows;
    q->kp.part_fused = false;
    const uint64_t tw = uint64_t(partition_tuple_u32_words(q->kp)) / 2;
    q->kp.part_fused = q->kp.part_bits > 8 && !q->part_fused_off &&
                       (end - begin) * tw * 8 <= (16ull << 30);
  }
  q->source = generate_kernel_source(q->kp);
  Status st = compile_kernel(ctx, q->source, &q->module, true);
  if (!st.ok()) return st;
  // A plan with many columns / state words can outgrow the 128 VGPRs a 1024-thread
  // workgroup leaves each wave: the scan kernel then keeps part of a tile in scratch
  // memory.  Fewer unroll steps per tile (fewer loads in flight, no scratch) are tried
  // until the kernel fits.
  while (q->kp.unroll > 1) {
    // (the kernels that hold a tile in registers: the fused scan and, for partitioned
    // plans, count and scatter -- only the latter run then)
    int scratch = 0;
    hipFunction_t fns[3] = {q->kp.partitioned ? nullptr : q->module.fn, q->module.fn_count,
                            q->module.fn_scatter};
    for (hipFunction_t f : fns) {
      int sc = 0;
      if (f && hipFuncGetAttribute(&sc, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, f) == hipSuccess) {
        scratch = std::max(scratch, sc);
      }
    }
    if (scratch == 0) break;
    q->kp.unroll /= 2;
    q->source = generate_kernel_source(q->kp);
    st = compile_kernel(ctx, q->source, &q->module, true);
    if (!st.ok()) return st;
  }
  // persistent grid: one wave of workgroups per CU slot
  int per_cu = 1;
  hipError_t oe = hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, q->module.fn,
                                                                     q->kp.block, 0);
  if (oe != hipSuccess || per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  q->grid = ctx->num_cus * per_cu;
  return Status();
}

// Plans without a cardinality hint (the reference's planner has none): before the
// first full run the fused kernel aggregates a prefix of the scan range, and the
// number of groups it finds among the passing rows gives the total by the occupancy
// formula d = G (1 - exp(-p / G)).  A plan whose groups will not fit the LDS tables
// is re-shaped for the partitioned path (choose_launch_shape) in the same execute.
static const uint64_t kProbeMinRows = 8ull << 20;  // below this the probe cannot pay
static const uint64_t kProbeRows = 256ull << 10;
static const uint64_t kProbeSegments = 16;  // row ranges spread evenly over the scan range

static void beat(evql_query* q) {
  if (q->hb && q->hb(q->hb_user) != 0) q->hb_abort = true;
}

static Status probe_cardinality(evql_query* q) {
  evql_table* t = q->table;
  const uint64_t nrows = q->nested ? q->nested_rows : t->layout.num_rows;
  const uint64_t begin = std::min(q->row_begin, nrows);
  const uint64_t end = q->row_end ? std::min(q->row_end, nrows) : nrows;
  if (end - begin < kProbeMinRows) return Status();
  const uint64_t saved_begin = q->row_begin, saved_end = q->row_end;
  // room for one group per sampled row
  Status st = alloc_gtab(q, 4 * kProbeRows);
  if (!st.ok()) return st;
  // The sample is kProbeSegments row ranges spread over the whole scan range, all
  // aggregated into one table: a prefix alone misjudges tables whose keys follow the
  // row order (time-ordered partitions: a prefix of a sorted key column holds one group).
  const uint64_t seg_rows = kProbeRows / kProbeSegments;
  const uint64_t stride = (end - begin) / kProbeSegments;
  for (uint64_t sg = 0; sg < kProbeSegments && st.ok(); ++sg) {
    q->row_begin = begin + sg * stride;
    q->row_end = q->row_begin + seg_rows;
    q->keep_table = sg > 0;
    st = query_launch(q);
    if (st.ok()) st = query_finish(q);
    beat(q);
  }
  q->keep_table = false;
  q->row_begin = saved_begin;
  q->row_end = saved_end;
  if (!st.ok()) return st;  // (a division by zero in the sample is one in the whole scan)
  const double p = double(q->stats.rows_passed), d = double(q->stats.num_groups);
  // the group table is rebuilt for the real run
  hipFree(q->d_gtab);
  q->d_gtab = nullptr;
  q->gcap = 0;
  q->executed = false;
  if (p < 1 || d < 1) return Status();
  const double p_total = p * double(end - begin) / double(kProbeRows);
  double g_est;
  if (d >= 0.999 * p) {
    g_est = p_total;  // (nearly) every sampled row its own group
  } else {
    // solve d = G (1 - exp(-p / G)) for G >= d by bisection (monotone in G)
    double lo = d, hi = std::max(p_total, d) * 4 + 16;
    for (int i = 0; i < 80; ++i) {
      const double mid = 0.5 * (lo + hi);
      const double dm = mid * (1.0 - std::exp(-p / mid));
      if (dm < d) lo = mid; else hi = mid;
    }
    g_est = std::min(0.5 * (lo + hi), p_total);
  }
  const uint64_t hint = uint64_t(g_est * 1.25) + 16;  // headroom for the estimate's error
  q->groups_hint = hint;
  q->stats.estimated_groups = hint;
  if (hint > kPartitionAboveSlots * lds_table_max_slots(q->kp) && partitioned_path_possible(q->kp)) {
    choose_launch_shape(&q->kp, hint);
    return compile_plan_kernels(q);
  }
  if (d >= 2 && d <= 4 && p >= 4096) {
    // a handful of groups among thousands of sampled rows: the shape for 2 .. 4 groups
    // (four lane-private accumulators, choose_launch_shape)
    choose_launch_shape(&q->kp, uint64_t(d));
    if (q->kp.lane_cache > 1) return compile_plan_kernels(q);
  }
  return Status();
}

// the kernel arguments that do not change between launches of one operator
static void fill_host_args(evql_query* q, HostArgs* ap) {
  HostArgs& a = *ap;
  evql_table* t = q->table;
  const KernelPlan& kp = q->kp;
  a.image = t->d_image;
  const uint64_t nrows = q->nested ? q->nested_rows : t->layout.num_rows;
  a.row_begin = std::min(q->row_begin, nrows);
  a.row_end = q->row_end ? std::min(q->row_end, nrows) : nrows;
  const uint64_t T = uint64_t(kp.tile_rows());
  a.tile0 = a.row_begin / T;
  a.ntiles = a.row_end > a.row_begin ? (a.row_end + T - 1) / T - a.tile0 : 0;
  a.row_filter = q->d_row_filter;
  a.row_filter_len = q->row_filter_len;
  a.gtab = q->d_gtab;
  a.gcap = q->gcap;
  a.status = q->d_status;
  a.counters = q->d_counters;
  for (int k = 0; k < kp.n_exact; ++k) {
    a.fscale[k] = std::ldexp(1.0, -q->fsum_exp[k]);
    a.fbound[k] = q->fsum_bound[k];
  }
  for (size_t i = 0; i < kp.cols.size(); ++i) {
    const ColAccess& c = kp.cols[i];
    a.col[i].base = t->d_image;
    if (c.layout_index >= 0) {
      a.col[i].pages = t->d_pages[c.layout_index][0];
      a.col[i].npages = t->layout.columns[c.layout_index].data_pages.size();
    }
    if (c.dict_code) {
      const StringDict& d = t->dicts[c.name];
      a.col[i].pages = d.d_code_pages;
      a.col[i].base = reinterpret_cast<const uint8_t*>(d.d_codes);
    } else if (c.packed && q->nested) {
      a.col[i].pages = q->nested_packed[i].pages;
      a.col[i].base = q->nested_packed[i].base;
    } else if (c.packed) {
      const MaterializedColumn& m = t->materialized[c.name];
      a.col[i].pages = m.d_packed_pages;
      a.col[i].base = m.d_packed;
    }
    if (q->nested) {
      a.col[i].soa = q->nested_flat[i];
      if (i < q->nested_strpos.size()) a.col[i].strpos = q->nested_strpos[i];
    } else if (c.mode == ColAccess::SOA) {
      const MaterializedColumn& m = t->materialized[c.name];
      a.col[i].soa = m.d_values;
      a.col[i].tags = m.d_tags;
      a.col[i].strpos = m.d_strpos;
    }
  }
}

// CSTableScan::fetchNext keeps the values of shallower columns across the rows of one
// slot, but after a row that WHERE rejects it resets every column at or below the
// running select level without re-reading it (CSTableScan.cc:501-512).  Worked out per
// column C of repetition depth c: the rows of a slot of C read C's value, except that
// they read 0 from the second row on when the slot's FIRST row was rejected (the first
// row itself always sees the freshly fetched value).  Whether a first row is rejected
// depends only on fresh values and on columns shallower than c, so the depths are
// settled one after the other, shallowest first: predicate of every row over the
// columns as they stand (evql_where_rows), verdict of every slot's first row
// (k_slot_keep), masked copy of the depth's columns (k_mask_parent).  Pinned by the
// reference's own engine on tests/golden/ref_csql_nested.json.
static Status apply_where_resets(evql_query* q, const LeafLevels& leaf) {
  evql_table* t = q->table;
  hipStream_t s = q->ctx->stream;
  const KernelPlan& kp = q->kp;
  const uint64_t n = q->nested_rows;
  if (n == 0 || !leaf.levels || !q->module.fn_where) return Status();
  uint32_t leaf_depth = 0;
  std::vector<uint32_t> depths;
  for (const auto& c : kp.cols) {
    leaf_depth = std::max(leaf_depth, t->layout.columns[c.layout_index].rlevel_max);
  }
  for (const auto& c : kp.cols) {
    const uint32_t d = t->layout.columns[c.layout_index].rlevel_max;
    if (d < leaf_depth && std::find(depths.begin(), depths.end(), d) == depths.end()) depths.push_back(d);
  }
  std::sort(depths.begin(), depths.end());
  const uint64_t np = padded_rows(n);
  const uint64_t ntiles = (n + kDecodeTile - 1) / kDecodeTile;
  DevBuf<uint8_t> d_acc, d_keep;
  DevBuf<uint64_t> d_off;
  HIP_TRY(d_acc.alloc(np));
  HIP_TRY(d_keep.alloc(np));
  HIP_TRY(d_off.alloc((ntiles + 2) * 8));
  HIP_TRY(hipMemsetAsync(d_acc, 0, np, s));
  struct WhereArgs {
    HostArgs a;
    uint8_t* acc;
  };
  for (uint32_t d : depths) {
    WhereArgs wa{};
    fill_host_args(q, &wa.a);
    wa.acc = d_acc;
    size_t sz = sizeof(WhereArgs);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &wa, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz,
                      HIP_LAUNCH_PARAM_END};
    int grid = q->grid;
    if (uint64_t(grid) > wa.a.ntiles) grid = int(wa.a.ntiles);
    if (grid > 0) {
      HIP_TRY(hipModuleLaunchKernel(q->module.fn_where, grid, 1, 1, kp.block, 1, 1, 0, s, nullptr,
                                    config));
    }
    HIP_TRY(launch_level_tile_counts(leaf.levels, d, n, d_off, s));
    HIP_TRY(launch_exclusive_scan(d_off, ntiles, nullptr, s));
    HIP_TRY(hipMemsetAsync(d_keep, 0, np, s));
    HIP_TRY(launch_slot_keep(leaf.levels, d_off, d, n, d_acc, d_keep, s));
    std::map<uint64_t*, uint64_t*> done;  // (a column referenced twice shares one buffer)
    for (size_t i = 0; i < kp.cols.size(); ++i) {
      if (t->layout.columns[kp.cols[i].layout_index].rlevel_max != d) continue;
      auto hit = done.find(q->nested_flat[i]);
      if (hit != done.end()) {
        q->nested_flat[i] = hit->second;
        continue;
      }
      DevBuf<uint64_t> d_out;
      HIP_TRY(d_out.alloc(np * 8));
      HIP_TRY(hipMemsetAsync(d_out, 0, np * 8, s));
      if (kp.cols[i].string_hash) {
        // a reset string reads "" (an all-zero SValue): mask the positions, hash again
        DevBuf<uint64_t> d_sp;
        HIP_TRY(d_sp.alloc(np * 8));
        HIP_TRY(hipMemsetAsync(d_sp, 0, np * 8, s));
        HIP_TRY(launch_mask_parent(leaf.levels, d_off, d, n, d_keep, q->nested_strpos[i], d_sp, s));
        HIP_TRY(launch_string_hash(t->d_image, t->d_pages[kp.cols[i].layout_index][0], d_sp, n,
                                   d_out, s));
        q->nested_strpos[i] = d_sp;
        q->nested_owned.push_back(d_sp.release());
      } else {
        HIP_TRY(launch_mask_parent(leaf.levels, d_off, d, n, d_keep, q->nested_flat[i], d_out, s));
      }
      done[q->nested_flat[i]] = d_out;
      q->nested_flat[i] = d_out;
      q->nested_owned.push_back(d_out.release());
    }
    HIP_TRY(hipStreamSynchronize(s));
  }
  return Status();
}

Status query_launch(evql_query* q) {
  evql_ctx* ctx = q->ctx;
  if (!q->probed && q->groups_hint == 0 && q->kp.key_mode != KEY_NONE && !q->within_record) {
    q->probed = true;
    Status st = probe_cardinality(q);
    if (!st.ok()) return st;
  }
  q->probed = true;
  q->merged = false;
  q->merged_dense = false;
  q->conv_valid = false;
  const KernelPlan& kp = q->kp;
  hipStream_t s = ctx->stream;
  if (!q->d_gtab) {
    // load factor <= 1/4 for small tables; very large ones (>= 1M groups) are kept
    // at <= 1/2: initialising and scanning the table is then a visible part of a
    // step (2.7 GB of slots for 1e7 groups at 1/4)
    const uint64_t slack = q->groups_hint >= (1ull << 20) ? 2 : 4;
    uint64_t want = kp.key_mode == KEY_NONE ? 8 : std::max<uint64_t>(q->groups_hint * slack, 1 << 16);
    // partitioned path: groups leave the LDS tables as dense records; the HBM table
    // only takes the groups of buckets that overflowed theirs (regrown on demand)
    if (kp.partitioned) want = 1 << 16;
    uint64_t cap = 8;
    while (cap < want) cap <<= 1;
    Status st = alloc_gtab(q, cap);
    if (!st.ok()) return st;
  }
  const uint64_t stride = q->gcap + 8;
  if (!q->keep_table) {
    TableInitArgs ia{};
    ia.words = q->d_gtab;
    ia.stride = stride;
    ia.nwords = uint32_t(kp.words_per_slot());
    ia.identity[0] = 0xFFFFFFFFFFFFFFFFull;
    int w = 1;
    if (kp.has_ident2()) ia.identity[w++] = 0xFFFFFFFFFFFFFFFFull;
    if (kp.need_first_row) ia.identity[w++] = 0xFFFFFFFFFFFFFFFFull;
    for (const auto& sw : kp.states) ia.identity[w++] = word_identity(sw.op);
    HIP_TRY(launch_table_init(ia, s));
    HIP_TRY(hipMemsetAsync(q->d_status, 0, 16, s));
    HIP_TRY(hipMemsetAsync(q->d_counters, 0, 64, s));
  } else {
    HIP_TRY(hipMemsetAsync(q->d_counters + 4, 0, 8, s));  // (the group count is recounted)
  }

  HostArgs a{};
  fill_host_args(q, &a);
  if (kp.n_distinct > 0 && !q->keep_table) {
    // count_distinct pair sets: emptied before every launch
    if (q->pairset_cap == 0) {
      const uint64_t span = a.row_end > a.row_begin ? a.row_end - a.row_begin : 0;
      // starts at <= 2^20 triples; a full set is regrown x4 and the query re-run
      uint64_t cap = 1 << 16;
      while (cap < 2 * span && cap < (1ull << 20)) cap <<= 1;
      q->pairset_cap = cap;
    }
    for (int i = 0; i < kp.n_distinct; ++i) {
      if (!q->d_pairset[i]) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_pairset[i]), q->pairset_cap * 3 * 8));
      }
      HIP_TRY(hipMemsetAsync(q->d_pairset[i], 0xff, q->pairset_cap * 3 * 8, s));
    }
  }
  for (int i = 0; i < kp.n_distinct; ++i) {
    a.pairset[i] = q->d_pairset[i];
    a.pairset_cap[i] = q->pairset_cap;
  }
  if (kp.partitioned && a.ntiles > 0) {
    // count -> per-bucket prefix -> scatter -> per-bucket LDS aggregation
    const uint64_t npart = 1ull << kp.part_bits;
    const uint64_t nwg = std::min<uint64_t>(uint64_t(q->grid), a.ntiles);
    HostArgsWithPart ap{};
    ap.a = a;
    ap.p.tiles_per_wg = (a.ntiles + nwg - 1) / nwg;
    ap.p.nwg = nwg;
    const bool two_level = q->module.fn_refine != nullptr;
    const uint64_t ncursors = 256 + npart;  // [coarse] + [fine]
    if (!q->d_part_counts) {
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_part_counts),
                        npart * uint64_t(q->grid) * sizeof(uint32_t)));
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_bucket_start), (npart + 2) * 8));
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_part_cursors), ncursors * 4));
    }
    ap.p.counts = q->d_part_counts;
    ap.p.bucket_start = q->d_bucket_start;
    ap.p.tuples = q->d_tuples;
    ap.p.tuples_tmp = q->d_tuples_tmp;
    ap.p.cursors = q->d_part_cursors;
    if (!q->d_dense) {
      q->dense_cap = std::max<uint64_t>(q->groups_hint, 1) * 2 + 4096;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_dense),
                        q->dense_cap * uint64_t(kp.words_per_slot() + 1) * 8));
    }
    ap.p.dense = q->d_dense;
    ap.p.dense_cap = q->dense_cap;
    q->dense_n = 0;
    HIP_TRY(hipMemsetAsync(q->d_part_cursors, 0, ncursors * 4, s));
    size_t psz = sizeof(HostArgsWithPart);
    void* pconfig[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ap, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psz,
                       HIP_LAUNCH_PARAM_END};
    const uint64_t tw = uint64_t(partition_tuple_u32_words(kp)) / 2;  // 8-byte words
    const uint64_t span = a.row_end - a.row_begin;
    uint64_t* d_total = q->d_bucket_start + npart + 1;
    const bool fused = kp.part_fused && two_level;
    HIP_TRY(hipEventRecord(q->ev0, s));
    HIP_TRY(hipMemsetAsync(q->d_bucket_start, 0, (npart + 2) * 8, s));
    uint64_t ntuples = span;  // upper bound: every row passes
    if (!fused) {
      HIP_TRY(hipModuleLaunchKernel(q->module.fn_count, unsigned(nwg), 1, 1, kp.block, 1, 1, 0, s,
                                    nullptr, pconfig));
      HIP_TRY(launch_part_scan(q->d_part_counts, npart, nwg, q->d_bucket_start, s));
      HIP_TRY(launch_exclusive_scan(q->d_bucket_start, npart + 1, d_total, s));
      if (q->tuples_cap < span && span * tw * 8 > (16ull << 30)) {
        // large scans with a selective predicate: size the buffers by the count pass
        HIP_TRY(hipMemcpyAsync(&ntuples, d_total, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
      }
    }
    if (ntuples > q->tuples_cap) {
      if (q->d_tuples) hipFree(q->d_tuples);
      q->d_tuples = nullptr;
      const uint64_t cap = ntuples + ntuples / 16 + 1024;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_tuples), cap * tw * 8));
      q->tuples_cap = cap;
    }
    if (two_level) {
      // coarse-bucket order.  Fused form: every coarse bucket owns a fixed range with 5 %
      // slack over an even share of the rows (+ one tile's worth); the hash spreads the
      // tuples evenly (64 buckets of ~2e6 tuples deviate by ~0.1 %), a bucket that
      // overflows anyway (one dominant key) voids the launch: exact offsets then
      const uint64_t ncoarse = 64;
      ap.p.coarse_cap = span / ncoarse + span / (ncoarse * 20) + 16384;
      const uint64_t want = fused ? ap.p.coarse_cap * ncoarse : q->tuples_cap;
      if (want > q->tuples_tmp_cap) {
        if (q->d_tuples_tmp) hipFree(q->d_tuples_tmp);
        q->d_tuples_tmp = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_tuples_tmp), want * tw * 8));
        q->tuples_tmp_cap = want;
      }
    }
    ap.p.tuples = q->d_tuples;
    ap.p.tuples_tmp = q->d_tuples_tmp;
    HIP_TRY(hipModuleLaunchKernel(q->module.fn_scatter, unsigned(nwg), 1, 1, kp.block, 1, 1, 0, s,
                                  nullptr, pconfig));
    if (fused) {
      // the fine-bucket sizes the scatter counted -> bucket_start[]
      HIP_TRY(launch_exclusive_scan(q->d_bucket_start, npart + 1, d_total, s));
    }
    // two workgroups per CU where the resources allow it: both passes wait on
    // dependent loads (tuple -> slot) and hide each other's latency
    const uint64_t wide = std::max<uint64_t>(uint64_t(q->grid), uint64_t(ctx->num_cus) * 2);
    if (two_level) {
      HIP_TRY(hipModuleLaunchKernel(q->module.fn_refine, unsigned(wide), 1, 1, kp.block, 1, 1, 0, s,
                                    nullptr, pconfig));
    }
    const unsigned agrid = unsigned(std::min<uint64_t>(npart, wide));
    HIP_TRY(hipModuleLaunchKernel(q->module.fn_aggregate, agrid, 1, 1, kp.block, 1, 1, 0, s, nullptr,
                                  pconfig));
    HIP_TRY(hipEventRecord(q->ev1, s));
    // group count into counter word 4 (read back by finish together with the rest)
    HIP_TRY(launch_table_compact(q->d_gtab, q->gcap, q->gcap + 8, uint32_t(kp.words_per_slot()),
                                 nullptr, 0, q->d_counters + 4, s));
    q->launched = true;
    q->stats.n_kernel_launches = 7;
    q->stats.rows_scanned = a.row_end - a.row_begin;
    return Status();
  }
  size_t sz = sizeof(HostArgs);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz,
                    HIP_LAUNCH_PARAM_END};
  HIP_TRY(hipEventRecord(q->ev0, s));
  if (a.ntiles > 0) {
    int grid = q->grid;
    if (uint64_t(grid) > a.ntiles) grid = int(a.ntiles);
    HIP_TRY(hipModuleLaunchKernel(q->module.fn, grid, 1, 1, kp.block, 1, 1, 0, s, nullptr, config));
  }
  HIP_TRY(hipEventRecord(q->ev1, s));
  // group count into counter word 4 (read back by finish together with the rest)
  HIP_TRY(launch_table_compact(q->d_gtab, q->gcap, q->gcap + 8, uint32_t(kp.words_per_slot()),
                               nullptr, 0, q->d_counters + 4, s));
  q->launched = true;
  q->stats.n_kernel_launches = 3;
  q->stats.rows_scanned = a.row_end - a.row_begin;
  return Status();
}

static Status fetch_results(evql_query* q);

// The groups of `q` as dense records of rplan()'s layout.  Plans that ran on dictionary
// codes are translated here, once per execute and only when somebody asks: code ->
// the hashed identity words of the string + its first row (k_dict_records).
Status query_records_view(evql_query* q, RecordsView* v) {
  if (!q->dict_key) {
    v->dense = q->d_dense;
    v->nd = std::min(q->dense_n, q->ngroups);
    return Status();
  }
  hipStream_t s = q->ctx->stream;
  const uint64_t n = q->ngroups;
  if (!q->conv_valid && n) {
    const uint32_t in_words = uint32_t(q->kp.words_per_slot()) + 1;
    const uint64_t nd = std::min(q->dense_n, n);
    DevBuf<uint64_t> tmp;
    const uint64_t* src = q->d_dense;
    if (n > nd) {  // groups of overflowed buckets / of the LDS path sit in the HBM table
      HIP_TRY(tmp.alloc(n * in_words * 8));
      if (nd) HIP_TRY(hipMemcpyAsync(tmp, q->d_dense, nd * in_words * 8, hipMemcpyDeviceToDevice, s));
      uint64_t* d_cnt = q->d_counters + 6;
      HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, s));
      HIP_TRY(launch_table_compact(q->d_gtab, q->gcap, q->gcap + 8, in_words - 1,
                                   tmp.p + nd * in_words, n - nd, d_cnt, s));
      src = tmp;
    }
    if (q->conv_cap < n) {
      if (q->d_conv) hipFree(q->d_conv);
      q->d_conv = nullptr;
      q->conv_cap = n + n / 8 + 1024;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_conv), q->conv_cap * uint64_t(in_words + 2) * 8));
    }
    const StringDict& d = q->table->dicts[q->rkp.cols[q->dict_candidate].name];
    HIP_TRY(launch_dict_records(src, n, in_words, d.d_entries, q->d_conv, s));
    HIP_TRY(hipStreamSynchronize(s));  // (tmp lives until here)
    q->conv_valid = true;
  }
  v->dense = q->d_conv;
  v->nd = n;
  return Status();
}

Status query_finish(evql_query* q) {
  if (!q->launched) return Status::error(EVQL_EARG, "query was not launched");
  evql_ctx* ctx = q->ctx;
  for (int attempt = 0; attempt < 12; ++attempt) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    uint32_t status[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(status, q->d_status, 16, hipMemcpyDeviceToHost));
    if (status[0] & 1u) return Status::error(EVQL_ERUNTIME, "division by zero");
    if (status[0] & 4u) return Status::error(EVQL_ERUNTIME, "modulo by zero");
    if (status[0] & 32u) {
      return Status::error(EVQL_ERUNTIME, "exact float sum: a value is not finite or exceeds the bound");
    }
    if (status[0] & 64u) {
      // a coarse bucket outgrew its slack (skewed keys): exact offsets from the count pass
      q->part_fused_off = true;
      Status stc = compile_plan_kernels(q);
      if (!stc.ok()) return stc;
      beat(q);
      Status st = query_launch(q);
      if (!st.ok()) return st;
      continue;
    }
    if (status[0] & (2u | 8u | 16u)) {
      // group table / count_distinct pair set / dense record buffer too small: grow
      // and run again
      Status st;
      beat(q);
      if (status[0] & 16u) {
        hipFree(q->d_dense);
        q->d_dense = nullptr;
        q->groups_hint = std::max<uint64_t>(q->groups_hint, 1024) * 4;
        if (q->kp.partitioned && !q->keep_table) {
          // the estimate was off by more than the headroom: bucket bits and launch shape
          // follow the corrected group count (too few buckets overflow every LDS table)
          const int old_bits = q->kp.part_bits;
          choose_launch_shape(&q->kp, q->groups_hint);
          if (q->kp.part_bits != old_bits || !q->kp.partitioned) {
            for (void* p : {(void*) q->d_part_counts, (void*) q->d_bucket_start, (void*) q->d_part_cursors}) {
              if (p) hipFree(p);
            }
            q->d_part_counts = nullptr;
            q->d_bucket_start = nullptr;
            q->d_part_cursors = nullptr;
            Status stc = compile_plan_kernels(q);
            if (!stc.ok()) return stc;
          }
        }
      }
      if (status[0] & 2u) {
        st = alloc_gtab(q, q->gcap * 4);
        if (!st.ok()) return st;
      }
      if (status[0] & 8u) {
        for (auto& p : q->d_pairset) {
          if (p) hipFree(p);
          p = nullptr;
        }
        q->pairset_cap *= 4;
      }
      st = query_launch(q);
      if (!st.ok()) return st;
      continue;
    }
    float ms = 0;
    hipEventElapsedTime(&ms, q->ev0, q->ev1);
    q->stats.kernel_ms = ms;
    // (a record scan's per-record reduction ran when the operator was built)
    q->stats.total_ms = ms + q->within_record_ms;
    uint64_t counters[8];
    HIP_TRY(hipMemcpy(counters, q->d_counters, 64, hipMemcpyDeviceToHost));
    q->stats.rows_passed = counters[0];
    q->stats.used_lds_table = q->kp.lds_slots > 0;
    q->launched = false;
    // the groups stay in HBM; they are compacted and copied to the host only
    // when the first nextBatch asks for them (a partial aggregate that is merged
    // on the device never leaves it).  Only the group count is read back: the
    // count pass was enqueued behind the kernels by launch (counter word 4).
    q->dense_n = q->kp.partitioned ? counters[3] : 0;
    q->ngroups = counters[4] + q->dense_n;
    q->stats.num_groups = q->ngroups;
    q->executed = true;
    q->fetched = false;
    q->emit_pos = 0;
    return Status();
  }
  return Status::error(EVQL_ENOMEM, "group table kept overflowing");
}

// number of occupied slots after the table was changed behind the host's back
// (import of another partition's groups)
Status query_recount(evql_query* q) {
  evql_ctx* ctx = q->ctx;
  uint64_t* d_cnt = q->d_counters + 4;
  HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, ctx->stream));
  HIP_TRY(launch_table_compact(q->d_gtab, q->gcap, q->gcap + 8, uint32_t(q->kp.words_per_slot()),
                               nullptr, 0, d_cnt, ctx->stream));
  uint64_t n = 0;
  HIP_TRY(hipMemcpyAsync(&n, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  q->ngroups = n + q->dense_n;
  q->stats.num_groups = q->ngroups;
  q->fetched = false;
  q->executed = true;
  q->emit_pos = 0;
  return Status();
}

// Moves every group of the query -- slots of the HBM hash table and the dense records of
// the partitioned path -- into a fresh hash table with room for `total` groups at load
// factor <= 1/2.
static Status rebuild_table(evql_query* q, uint64_t total) {
  evql_ctx* ctx = q->ctx;
  hipStream_t s = ctx->stream;
  const KernelPlan& kp = q->kp;
  const uint32_t nwords = uint32_t(kp.words_per_slot());
  // groups already in the table (overflowed buckets of the partitioned path, or all)
  uint64_t in_table = q->ngroups - q->dense_n;
  DevBuf<uint64_t> old_rec;
  if (in_table) {
    HIP_TRY(old_rec.alloc(in_table * (nwords + 1) * 8));
    uint64_t* d_cnt = q->d_counters + 6;
    HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, s));
    HIP_TRY(launch_table_compact(q->d_gtab, q->gcap, q->gcap + 8, nwords, old_rec, in_table, d_cnt, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  uint64_t cap = 1 << 16;
  while (cap < total * 2) cap <<= 1;
  Status st = alloc_gtab(q, cap);
  if (!st.ok()) return st;
  TableInitArgs ia{};
  ia.words = q->d_gtab;
  ia.stride = q->gcap + 8;
  ia.nwords = nwords;
  ia.identity[0] = 0xFFFFFFFFFFFFFFFFull;
  int w = 1;
  if (kp.has_ident2()) ia.identity[w++] = 0xFFFFFFFFFFFFFFFFull;
  if (kp.need_first_row) ia.identity[w++] = 0xFFFFFFFFFFFFFFFFull;
  for (const auto& sw : kp.states) ia.identity[w++] = word_identity(sw.op);
  HIP_TRY(launch_table_init(ia, s));
  MergeArgs a{};
  a.words = q->d_gtab;
  a.gcap = q->gcap;
  a.stride = q->gcap + 8;
  a.nwords = nwords;
  w = 1;
  a.has_ident2 = kp.has_ident2() ? 1 : 0;
  if (kp.has_ident2()) a.ops[w++] = 255;
  if (kp.need_first_row) a.ops[w++] = 2;  // min
  for (const auto& sw : kp.states) a.ops[w++] = uint32_t(sw.op);
  a.status = q->d_status;
  HIP_TRY(hipMemsetAsync(q->d_status, 0, 16, s));
  if (in_table) HIP_TRY(launch_table_merge(a, old_rec, in_table, s));
  if (q->dense_n) HIP_TRY(launch_table_merge(a, q->d_dense, q->dense_n, s));
  uint32_t status[4] = {0};
  HIP_TRY(hipMemcpyAsync(status, q->d_status, 16, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (status[0] & 2u) return Status::error(EVQL_ENOMEM, "group table full");
  q->dense_n = 0;
  return Status();
}

// the dense records of the partitioned path moved into the (regrown) HBM hash table:
// what merging another partition's groups into this query needs
Status query_dense_into_table(evql_query* q) {
  if (q->dense_n == 0) return Status();
  return rebuild_table(q, q->ngroups);
}

// room for `extra` more groups: the table is rebuilt BEFORE a merge could fill it, so a
// merge never stops half way (GroupByMergeExpression's map simply grows, groupby.cc:528-637)
Status query_reserve_groups(evql_query* q, uint64_t extra) {
  if (q->dense_n == 0 && (q->ngroups + extra) * 2 <= q->gcap) return Status();
  return rebuild_table(q, q->ngroups + extra);
}

// count_distinct pairs of another partition into this query's set (aggregate.cc:119-137:
// mergeInstance inserts the other set's values); every pair that is new adds 1 to its
// group's aggregate.  The set is regrown first when the pairs might not fit.
Status query_import_pairs(evql_query* q, int which, const uint64_t* d_triples, uint64_t n) {
  hipStream_t s = q->ctx->stream;
  const KernelPlan& kp = q->kp;
  if (q->dense_n) {
    Status st = query_dense_into_table(q);
    if (!st.ok()) return st;
  }
  // pairs held today (all sets share one capacity: the scan kernel takes one)
  uint64_t held_max = 0;
  std::vector<uint64_t> held(kp.n_distinct, 0);
  for (int d = 0; d < kp.n_distinct; ++d) {
    if (!q->d_pairset[d]) continue;
    uint64_t* d_cnt = q->d_counters + 6;
    HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, s));
    HIP_TRY(launch_pairset_export(q->d_pairset[d], q->pairset_cap, nullptr, 0, d_cnt, s));
    HIP_TRY(hipMemcpyAsync(&held[d], d_cnt, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    held_max = std::max(held_max, held[d]);
  }
  const uint64_t need = std::max(held_max, held[which] + n);
  uint64_t cap = q->pairset_cap ? q->pairset_cap : (1 << 16);
  while (cap < need * 2) cap <<= 1;
  if (cap != q->pairset_cap) {
    // regrow every set: stored triples re-inserted as they are, nothing counted again
    for (int d = 0; d < kp.n_distinct; ++d) {
      DevBuf<uint64_t> d_new, d_tr;
      HIP_TRY(d_new.alloc(cap * 24));
      HIP_TRY(hipMemsetAsync(d_new, 0xff, cap * 24, s));
      if (q->d_pairset[d] && held[d]) {
        HIP_TRY(d_tr.alloc(held[d] * 24));
        uint64_t* d_cnt = q->d_counters + 6;
        HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, s));
        HIP_TRY(launch_pairset_export(q->d_pairset[d], q->pairset_cap, d_tr, held[d], d_cnt, s));
        PairsetMergeArgs pa{};
        pa.set = d_new;
        pa.set_cap = cap;
        pa.words = nullptr;
        pa.status = q->d_status;
        HIP_TRY(launch_pairset_merge(pa, d_tr, held[d], s));
        HIP_TRY(hipStreamSynchronize(s));
      }
      if (q->d_pairset[d]) hipFree(q->d_pairset[d]);
      q->d_pairset[d] = d_new.release();
    }
    q->pairset_cap = cap;
  }
  for (int d = 0; d < kp.n_distinct; ++d) {
    if (q->d_pairset[d]) continue;  // (an empty merge target: evql_query_reset)
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_pairset[d]), q->pairset_cap * 24));
    HIP_TRY(hipMemsetAsync(q->d_pairset[d], 0xff, q->pairset_cap * 24, s));
  }
  if (n == 0) return Status();
  int word = -1;
  for (const auto& ag : kp.aggs) {
    if (ag.distinct_index == which) word = kp.state_word_base() + ag.first_word;
  }
  PairsetMergeArgs pa{};
  pa.set = q->d_pairset[which];
  pa.set_cap = q->pairset_cap;
  pa.words = q->d_gtab;
  pa.gcap = q->gcap;
  pa.nwords = uint32_t(kp.words_per_slot());
  pa.word = uint32_t(word);
  pa.key_mode = uint32_t(kp.key_mode);
  pa.status = q->d_status;
  HIP_TRY(hipMemsetAsync(q->d_status, 0, 16, s));
  HIP_TRY(launch_pairset_merge(pa, d_triples, n, s));
  uint32_t status[4] = {0};
  HIP_TRY(hipMemcpyAsync(status, q->d_status, 16, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (status[0] & 8u) return Status::error(EVQL_ENOMEM, "count_distinct set full");
  if (status[0] & 2u) {
    return Status::error(EVQL_EARG, "a pair's group is not in the table: import the group records first");
  }
  q->fetched = false;
  return Status();
}

// (re)creates an empty group table without scanning: the merge target of
// GroupByMergeExpression (groupby.cc:528-637)
Status query_reset(evql_query* q) {
  evql_ctx* ctx = q->ctx;
  const KernelPlan& kp = q->kp;
  if (!q->d_gtab) {
    uint64_t want = kp.key_mode == KEY_NONE ? 8 : std::max<uint64_t>(q->groups_hint * 4, 1 << 16);
    uint64_t cap = 8;
    while (cap < want) cap <<= 1;
    Status st = alloc_gtab(q, cap);
    if (!st.ok()) return st;
  }
  TableInitArgs ia{};
  ia.words = q->d_gtab;
  ia.stride = q->gcap + 8;
  ia.nwords = uint32_t(kp.words_per_slot());
  ia.identity[0] = 0xFFFFFFFFFFFFFFFFull;
  int w = 1;
  if (kp.has_ident2()) ia.identity[w++] = 0xFFFFFFFFFFFFFFFFull;
  if (kp.need_first_row) ia.identity[w++] = 0xFFFFFFFFFFFFFFFFull;
  for (const auto& sw : kp.states) ia.identity[w++] = word_identity(sw.op);
  HIP_TRY(launch_table_init(ia, ctx->stream));
  HIP_TRY(hipMemsetAsync(q->d_status, 0, 16, ctx->stream));
  HIP_TRY(hipMemsetAsync(q->d_counters, 0, 64, ctx->stream));
  HIP_TRY(hipEventRecord(q->ev0, ctx->stream));
  HIP_TRY(hipEventRecord(q->ev1, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  q->ngroups = 0;
  q->dense_n = 0;
  q->merged = false;
  q->merged_dense = false;
  q->stats.num_groups = 0;
  q->stats.rows_scanned = 0;
  q->stats.rows_passed = 0;
  q->executed = true;
  q->fetched = false;
  q->launched = false;
  q->emit_pos = 0;
  return Status();
}

// ---------------------------------------------------------------------------
// large results: the output columns packed on the device
// ---------------------------------------------------------------------------
// GroupByExpression::nextBatch (groupby.cc:187-220) runs `method_call` of every select
// expression per group and appends the value to the column's SVector.  For 1e7 groups the
// host loop (record copy, sort, one eval_expr per cell) took seconds.  When every select
// expression is the group key, a bare aggregate or the first-row value of a scan column --
// config 4 / 4s, and what `select k, count(1), sum(x) .. group by k` looks like -- one
// kernel writes the packed SVector bytes (svalue.cc:410-517) of every column for ALL groups;
// the host copies each column once into pinned memory and next_batch hands out slices.
// Row order is unspecified for a GROUP BY; small results keep the host path
// and its deterministic order.
static const uint64_t kDeviceEmitMinGroups = 1 << 16;

static bool device_emit_columns(const evql_query* q, bool merged, EmitArgs* ea) {
  const KernelPlan& kp = q->rplan();
  const size_t nsel = q->select.size();
  if (q->group_mode != EVQL_MODE_FINAL || !q->order.empty() || q->has_limit) return false;
  if (nsel == 0 || nsel > kMaxEmitCols) return false;
  for (size_t i = 0; i < nsel; ++i) {
    const LoweredProgram& lp = q->select[i];
    EmitCol& e = ea->col[i];
    e = EmitCol{};
    e.count_word = -1;
    e.stype = lp.return_type;
    e.elem = lp.return_type == EVQL_T_BOOL ? 2 : 9;
    if (lp.return_type == EVQL_T_NIL) return false;
    if (lp.is_aggregate) {
      if (lp.call->kind != Expr::AGG_GET) return false;  // post-aggregate arithmetic: host
      const AggPlan& a = kp.aggs[q->select_agg_index[i]];
      if (a.exact_index >= 0) return false;  // (128-bit rounding of an exact sum: host)
      e.kind = 1;
      e.word = uint32_t(1 + kp.state_word_base() + a.first_word);
      switch (a.fn) {
        case EVQL_AGG_COUNT: case EVQL_AGG_SUM_UINT64: case EVQL_AGG_SUM_INT64:
        case EVQL_AGG_SUM_FLOAT64: case EVQL_AGG_COUNT_DISTINCT_UINT64:
          break;
        case EVQL_AGG_MEAN_UINT64: case EVQL_AGG_MEAN_INT64: case EVQL_AGG_MEAN_FLOAT64:
          e.is_mean = 1;
          e.count_word = int32_t(e.word + 1);
          break;
        default:  // min / max
          e.count_word = int32_t(e.word + 1);
      }
    } else if (q->select_passthrough[i]) {
      if (lp.return_type == EVQL_T_STRING) return false;
      e.kind = 0;
    } else {
      // a bare column: select expr = X_INPUT(j) of the scan select list = X_INPUT(c)
      if (merged || !kp.need_first_row || lp.call->kind != Expr::INPUT) return false;
      const uint32_t j = lp.call->input;
      if (j >= q->scan_select.size() || q->scan_select[j].call->kind != Expr::INPUT) return false;
      const uint32_t c = q->scan_select[j].call->input;
      if (c >= kp.cols.size() || q->nested) return false;
      const ColAccess& ca = kp.cols[c];
      if (ca.stype != lp.return_type) return false;
      e.kind = 2;
      e.src = c;
      e.to_float = ca.stype == EVQL_T_FLOAT64 && ca.from_uint_to_float;
      if (ca.string_hash) e.elem = 0;
    }
  }
  ea->ncols = uint32_t(nsel);
  return true;
}

static Status pinned_reserve(uint8_t** p, size_t* cap, size_t bytes) {
  if (bytes <= *cap && *p) return Status();
  if (*p) hipHostFree(*p);
  *p = nullptr;
  *cap = 0;
  const size_t want = bytes + bytes / 8 + 4096;
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(p), want, hipHostMallocDefault));
  *cap = want;
  return Status();
}

static Status emit_on_device(evql_query* q, EmitArgs& ea, const uint64_t* d_rec, uint64_t n,
                             uint32_t nwords) {
  evql_table* t = q->table;
  const KernelPlan& kp = q->rplan();
  hipStream_t s = q->ctx->stream;
  const uint32_t nsel = ea.ncols;
  evql_query::DeviceEmit& de = q->demit;
  de.col.resize(nsel, nullptr);
  de.col_cap.resize(nsel, 0);
  de.off.resize(nsel, nullptr);
  de.off_cap.resize(nsel, 0);
  de.elem.assign(nsel, 0);
  ea.records = d_rec;
  ea.n = n;
  ea.rw = nwords + 1;
  ea.image = t->d_image;
  bool need_first = false;
  for (uint32_t i = 0; i < nsel; ++i) need_first = need_first || ea.col[i].kind == 2;
  DevBuf<uint64_t> d_rows, d_vals;
  DevBuf<uint8_t> d_tags;
  DevBuf<RtColumn> d_cols;
  const uint32_t nc = uint32_t(kp.cols.size());
  if (need_first) {
    // the column values of every group's first row (strings: their strpos words)
    std::vector<RtColumn> rc(nc);
    for (uint32_t c = 0; c < nc; ++c) {
      const ColAccess& ca = kp.cols[c];
      rc[c] = RtColumn{};
      rc[c].pages = ca.layout_index >= 0 ? t->d_pages[ca.layout_index][0] : nullptr;
      rc[c].mode = ca.mode;
      rc[c].bits = ca.bits;
      if (ca.packed) {
        const MaterializedColumn& m = t->materialized[ca.name];
        rc[c].pages = m.d_packed_pages;
        rc[c].base = m.d_packed;
      } else if (ca.mode == ColAccess::SOA) {
        const MaterializedColumn& m = t->materialized[ca.name];
        rc[c].soa = ca.string_hash ? m.d_strpos : m.d_values;
        rc[c].tags = m.d_tags;
      }
    }
    HIP_TRY(d_rows.alloc(n * 8));
    HIP_TRY(d_cols.alloc(nc * sizeof(RtColumn)));
    HIP_TRY(d_vals.alloc(n * nc * 8));
    HIP_TRY(d_tags.alloc(n * nc));
    HIP_TRY(launch_extract_word(d_rec, n, nwords + 1, uint32_t(1 + kp.first_row_word()), d_rows, s));
    HIP_TRY(hipMemcpyAsync(d_cols, rc.data(), nc * sizeof(RtColumn), hipMemcpyHostToDevice, s));
    HIP_TRY(launch_gather_rows(t->d_image, d_cols, nc, d_rows, n, d_vals, d_tags, s));
    HIP_TRY(hipStreamSynchronize(s));  // (rc lives until here)
    ea.first_vals = d_vals;
    ea.first_tags = d_tags;
  }
  // device buffers of the packed columns
  std::vector<DevBuf<uint8_t>> d_out(nsel);
  std::vector<DevBuf<uint64_t>> d_off(nsel);
  std::vector<uint64_t> str_bytes(nsel, 0);
  DevBuf<EmitArgs> d_args;
  HIP_TRY(d_args.alloc(sizeof(EmitArgs)));
  for (uint32_t i = 0; i < nsel; ++i) {
    EmitCol& e = ea.col[i];
    de.elem[i] = e.elem;
    if (e.elem) {
      HIP_TRY(d_out[i].alloc(n * e.elem));
      e.out = d_out[i];
    } else {
      e.pages = t->d_pages[kp.cols[e.src].layout_index][0];
      HIP_TRY(d_off[i].alloc((n + 2) * 8));
    }
  }
  HIP_TRY(hipMemcpyAsync(d_args, &ea, sizeof(EmitArgs), hipMemcpyHostToDevice, s));
  HIP_TRY(launch_emit_fixed(d_args, n, s));
  bool strings = false;
  for (uint32_t i = 0; i < nsel; ++i) {
    if (ea.col[i].elem) continue;
    strings = true;
    HIP_TRY(launch_emit_str_sizes(d_args, i, n, d_off[i], s));
    HIP_TRY(launch_exclusive_scan(d_off[i], n, d_off[i].p + n, s));
    HIP_TRY(hipMemcpyAsync(&str_bytes[i], d_off[i].p + n, 8, hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  if (strings) {
    for (uint32_t i = 0; i < nsel; ++i) {
      if (ea.col[i].elem) continue;
      HIP_TRY(d_out[i].alloc(str_bytes[i] + 16));
      ea.col[i].out = d_out[i];
      ea.col[i].offsets = d_off[i];
    }
    HIP_TRY(hipMemcpyAsync(d_args, &ea, sizeof(EmitArgs), hipMemcpyHostToDevice, s));
    for (uint32_t i = 0; i < nsel; ++i) {
      if (!ea.col[i].elem) HIP_TRY(launch_emit_str_bytes(d_args, i, n, s));
    }
  }
  // one copy per column into pinned host memory
  for (uint32_t i = 0; i < nsel; ++i) {
    const size_t bytes = ea.col[i].elem ? size_t(n) * ea.col[i].elem : size_t(str_bytes[i]);
    Status st = pinned_reserve(&de.col[i], &de.col_cap[i], bytes);
    if (!st.ok()) return st;
    if (bytes) HIP_TRY(hipMemcpyAsync(de.col[i], d_out[i], bytes, hipMemcpyDeviceToHost, s));
    if (!ea.col[i].elem) {
      uint8_t* po = reinterpret_cast<uint8_t*>(de.off[i]);
      st = pinned_reserve(&po, &de.off_cap[i], (n + 1) * 8);
      de.off[i] = reinterpret_cast<uint64_t*>(po);
      if (!st.ok()) return st;
      HIP_TRY(hipMemcpyAsync(de.off[i], d_off[i], (n + 1) * 8, hipMemcpyDeviceToHost, s));
    }
  }
  HIP_TRY(hipStreamSynchronize(s));
  de.active = true;
  return Status();
}

static Status fetch_results(evql_query* q) {
  evql_ctx* ctx = q->ctx;
  evql_table* t = q->table;
  const KernelPlan& kp = q->rplan();
  hipStream_t s = ctx->stream;
  // after an exchange the groups live in the merged table (wider slots)
  const bool merged = q->merged;
  const uint32_t nwords = merged ? q->m_words : uint32_t(kp.words_per_slot());
  uint64_t* const gtab = merged ? q->d_mtab : q->d_gtab;
  const uint64_t gcap = merged ? (q->merged_dense ? q->mdense_n : q->mcap) : q->gcap;
  const uint64_t stride = gcap + 8;
  const uint64_t maxrec = gcap + 2;
  RecordsView view;
  if (!merged) {
    Status stv = query_records_view(q, &view);
    if (!stv.ok()) return stv;
  }
  const uint64_t dense_n = merged ? 0 : view.nd;
  uint64_t* d_rec = nullptr;
  uint64_t* d_cnt = q->d_counters + 5;
  // the record buffer is sized by the number of groups (counted by finish /
  // recount / reset), not by the table capacity
  uint64_t n = q->stats.num_groups;
  if (n > maxrec + dense_n) n = maxrec + dense_n;
  const uint64_t total_groups = n;
  // small results (the usual case) reuse a per-query 1 MiB buffer: no allocation
  // inside a step
  const size_t kSmallRec = 1 << 20;
  DevBuf<uint64_t> rec_own;  // records that do not fit the small buffer
  if (n && merged && q->merged_dense) {
    // (a bucketed merge left the groups as dense records already)
    n = std::min(n, q->mdense_n);
    d_rec = q->d_mdense;
  } else if (n) {
    if (n * (nwords + 1) * 8 <= kSmallRec) {
      if (!q->d_small_rec) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_small_rec), kSmallRec));
      d_rec = q->d_small_rec;
    } else {
      HIP_TRY(rec_own.alloc(n * (nwords + 1) * 8));
      d_rec = rec_own;
    }
    // dense records of the partitioned path first, the table's groups behind them
    const uint64_t nd = std::min(dense_n, n);
    if (nd) {
      HIP_TRY(hipMemcpyAsync(d_rec, view.dense, nd * (nwords + 1) * 8, hipMemcpyDeviceToDevice, s));
    }
    HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, s));
    if (n > nd) {
      HIP_TRY(launch_table_compact(gtab, gcap, stride, nwords, d_rec + nd * (nwords + 1),
                                   n - nd, d_cnt, s));
    }
  }
  // ORDER BY .. LIMIT: only the offset+limit smallest records by the first sort
  // key (plus, with further sort keys, every tie of the boundary key) leave the
  // device: radix select over the dense records, 8 bits per pass
  const uint64_t want = q->offset + q->limit;
  if (n && !q->order.empty() && q->has_limit && want < n) {
    uint64_t m = 0;
    if (want > 0) {
      DevBuf<uint64_t> d_keys, d_hist, d_idx, d_ctr;
      HIP_TRY(d_keys.alloc(n * 8));
      HIP_TRY(d_hist.alloc(256 * 8));
      HIP_TRY(d_idx.alloc(n * 8));
      HIP_TRY(d_ctr.alloc(3 * 8));
      OrderKeyArgs ka = q->order_key;
      ka.records = d_rec;
      ka.record_words = nwords + 1;
      ka.n = n;
      ka.keys = d_keys;
      HIP_TRY(launch_order_keys(ka, s));
      uint64_t hi_mask = 0, hi_value = 0, remaining = want;
      for (int shift = 56; shift >= 0; shift -= 8) {
        uint64_t hist[256];
        HIP_TRY(hipMemsetAsync(d_hist, 0, sizeof(hist), s));
        HIP_TRY(launch_radix_hist(d_keys, n, hi_mask, hi_value, uint32_t(shift), d_hist, s));
        HIP_TRY(hipMemcpyAsync(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        uint64_t d = 0;
        while (d < 255 && hist[d] < remaining) remaining -= hist[d++];
        hi_value |= d << shift;
        hi_mask |= 0xFFull << shift;
      }
      // `remaining` = how many records with key == hi_value the result needs
      const uint64_t max_eq = q->order.size() == 1 ? remaining : n;
      uint64_t ctr[3] = {0, 0, 0};
      HIP_TRY(hipMemsetAsync(d_ctr, 0, sizeof(ctr), s));
      HIP_TRY(launch_order_collect(d_keys, n, hi_value, max_eq, d_idx, d_ctr, s));
      HIP_TRY(hipMemcpyAsync(ctr, d_ctr, sizeof(ctr), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      m = ctr[2];
      DevBuf<uint64_t> d_rec2;
      HIP_TRY(d_rec2.alloc(m * (nwords + 1) * 8));
      HIP_TRY(launch_gather_records(d_rec, nwords + 1, d_idx, m, d_rec2, s));
      HIP_TRY(hipStreamSynchronize(s));
      rec_own.reset();
      rec_own.p = d_rec2.release();
      d_rec = rec_own;
    }
    n = m;
  }
  q->ngroups = n;
  q->rec_stride = nwords + 1;
  q->demit.active = false;
  if (n >= kDeviceEmitMinGroups) {
    EmitArgs ea{};
    if (device_emit_columns(q, merged, &ea)) {
      Status ste = emit_on_device(q, ea, d_rec, n, nwords);
      if (!ste.ok()) return ste;
      q->records.clear();
      q->distinct_values.clear();
      q->stats.num_groups = total_groups;
      q->emit_pos = 0;
      q->executed = true;
      q->fetched = true;
      return Status();
    }
  }
  q->records.assign(n * (nwords + 1), 0);
  if (n) {
    HIP_TRY(hipMemcpyAsync(q->records.data(), d_rec, n * (nwords + 1) * 8,
                           hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  // deterministic output order: by first row (scan order) or by identity
  {
    const size_t rw = nwords + 1;
    std::vector<uint64_t> idx(n);
    for (uint64_t i = 0; i < n; ++i) idx[i] = i;
    const size_t keyw = kp.need_first_row ? size_t(1 + kp.first_row_word()) : 1;
    const uint64_t* r = q->records.data();
    std::sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) {
      const uint64_t ka = r[a * rw + keyw], kb = r[b * rw + keyw];
      if (ka != kb) return ka < kb;
      return r[a * rw] < r[b * rw];
    });
    std::vector<uint64_t> sorted(q->records.size());
    for (uint64_t i = 0; i < n; ++i) {
      memcpy(&sorted[i * rw], &r[idx[i] * rw], rw * 8);
    }
    q->records.swap(sorted);
  }
  // first-row values of every scan column
  q->first_vals.clear();
  q->first_tags.clear();
  if (kp.need_first_row && n && merged) {
    // the merged records carry the first-row values themselves: one word per scan
    // column, one word of NULL-tag bits; string words point into the received bytes
    const uint32_t nc = uint32_t(kp.cols.size());
    const size_t rw = nwords + 1, fr0 = size_t(kp.words_per_slot()) + 1;
    q->first_vals.resize(n * nc);
    q->first_tags.resize(n * nc);
    q->first_str_off.assign(n * nc, 0);
    q->first_str_heap = q->m_heap;
    for (uint64_t i = 0; i < n; ++i) {
      const uint64_t* rec = &q->records[i * rw];
      const uint64_t tags = rec[fr0 + nc];
      for (uint32_t c = 0; c < nc; ++c) {
        q->first_vals[uint64_t(c) * n + i] = rec[fr0 + c];
        q->first_tags[uint64_t(c) * n + i] = uint8_t((tags >> c) & 1);
        if (kp.cols[c].string_hash) {
          const uint64_t off = rec[fr0 + c] & kStrOffMask;
          if (off + (rec[fr0 + c] >> 40) > q->first_str_heap.size()) {
            return Status::error(EVQL_ERUNTIME, "exchange: string offset outside the received bytes");
          }
          q->first_str_off[uint64_t(c) * n + i] = off;
        }
      }
    }
  } else if (kp.need_first_row && n) {
    const uint32_t nc = uint32_t(kp.cols.size());
    std::vector<uint64_t> rows(n);
    const size_t rw = nwords + 1;
    // (never hand an unrecorded first row to the gather: it would read far outside the table)
    const uint64_t row_limit = q->nested ? q->nested_rows : t->layout.num_rows;
    for (uint64_t i = 0; i < n; ++i) {
      rows[i] = q->records[i * rw + 1 + kp.first_row_word()];
      if (rows[i] >= row_limit) {
        return Status::error(EVQL_ERUNTIME, "a group's first row was not recorded");
      }
    }
    std::vector<RtColumn> rc(nc);
    for (uint32_t c = 0; c < nc; ++c) {
      const ColAccess& ca = kp.cols[c];
      rc[c].pages = ca.layout_index >= 0 ? t->d_pages[ca.layout_index][0] : nullptr;
      rc[c].mode = ca.mode;
      rc[c].bits = ca.bits;
      if (ca.packed && q->nested) {
        rc[c].pages = q->nested_packed[c].pages;
        rc[c].base = q->nested_packed[c].base;
      } else if (ca.packed) {
        const MaterializedColumn& m = t->materialized[ca.name];
        rc[c].pages = m.d_packed_pages;
        rc[c].base = m.d_packed;
      }
      if (q->nested) {
        rc[c].soa = ca.string_hash ? q->nested_strpos[c] : q->nested_flat[c];
      } else if (ca.mode == ColAccess::SOA) {
        const MaterializedColumn& m = t->materialized[ca.name];
        // strings: (len << 40 | position); their bytes are copied out below
        rc[c].soa = ca.string_hash ? m.d_strpos : m.d_values;
        rc[c].tags = m.d_tags;
      }
    }
    DevBuf<uint64_t> d_rows, d_vals;
    DevBuf<RtColumn> d_cols;
    DevBuf<uint8_t> d_tags;
    HIP_TRY(d_rows.alloc(n * 8));
    HIP_TRY(d_cols.alloc(nc * sizeof(RtColumn)));
    HIP_TRY(d_vals.alloc(n * nc * 8));
    HIP_TRY(d_tags.alloc(n * nc));
    HIP_TRY(hipMemcpyAsync(d_rows, rows.data(), n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_cols, rc.data(), nc * sizeof(RtColumn), hipMemcpyHostToDevice, s));
    HIP_TRY(launch_gather_rows(t->d_image, d_cols, nc, d_rows, n, d_vals, d_tags, s));
    q->first_vals.resize(n * nc);
    q->first_tags.resize(n * nc);
    HIP_TRY(hipMemcpyAsync(q->first_vals.data(), d_vals, n * nc * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(q->first_tags.data(), d_tags, n * nc, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // bytes of the first-row strings: one packed heap per string column
    q->first_str_off.assign(n * nc, 0);
    q->first_str_heap.clear();
    for (uint32_t c = 0; c < nc; ++c) {
      const ColAccess& ca = kp.cols[c];
      if (!ca.string_hash) continue;
      std::vector<uint64_t> offs(n);
      uint64_t total = q->first_str_heap.size();
      const uint64_t heap0 = total;
      for (uint64_t i = 0; i < n; ++i) {
        offs[i] = total - heap0;
        q->first_str_off[uint64_t(c) * n + i] = total;
        if (!q->first_tags[uint64_t(c) * n + i]) total += q->first_vals[uint64_t(c) * n + i] >> 40;
      }
      const uint64_t bytes = total - heap0;
      q->first_str_heap.resize(total);
      if (bytes == 0) continue;
      DevBuf<uint64_t> d_offs;
      DevBuf<uint8_t> d_heap;
      HIP_TRY(d_offs.alloc(n * 8));
      HIP_TRY(d_heap.alloc(bytes));
      HIP_TRY(hipMemcpyAsync(d_offs, offs.data(), n * 8, hipMemcpyHostToDevice, s));
      // (NULL rows carry strpos 0: length 0, nothing copied)
      HIP_TRY(launch_copy_strings(t->d_image, t->d_pages[ca.layout_index][0], d_vals.p + uint64_t(c) * n,
                                  d_offs, n, d_heap, s));
      HIP_TRY(hipMemcpyAsync(q->first_str_heap.data() + heap0, d_heap, bytes, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
    }
  }
  q->distinct_values.clear();
  if (q->group_mode == EVQL_MODE_PARTIAL && kp.n_distinct > 0) {
    // count_distinct's saved state is the set itself (aggregate.cc:111-117): the
    // (group, value, flags) triples of the aggregate's pair set, grouped on the host
    const bool hashed = kp.key_mode == KEY_HASHED;
    q->distinct_values.resize(kp.n_distinct);
    std::vector<uint64_t> host;
    for (int d = 0; d < kp.n_distinct; ++d) {
      // (merged results: the union of the ranks' / tables' sets, exchange.cc)
      const uint64_t cap = merged ? q->mset_cap[d] : q->pairset_cap;
      const uint64_t* d_set = merged ? q->d_mset[d] : q->d_pairset[d];
      host.assign(cap * 3, ~0ull);
      if (d_set && cap) HIP_TRY(hipMemcpy(host.data(), d_set, cap * 3 * 8, hipMemcpyDeviceToHost));
      auto& sets = q->distinct_values[d];
      for (uint64_t sl = 0; sl < cap; ++sl) {
        uint64_t ident = host[sl], value = host[cap + sl], flags = host[2 * cap + sl];
        if (ident == ~0ull || value == ~0ull || flags == ~0ull) continue;
        uint64_t second;
        if (hashed) {
          // (a value of 2^64-1 is stored as 2^64-2 with the flags word scrambled: such
          // a group is found under the unscrambled second identity word)
          second = flags;
          if (value == ~0ull - 1 && !sets.count({ident, second})) {
            const uint64_t alt = flags ^ 0xc2b2ae3d27d4eb4full;
            bool known = false;
            for (uint64_t g = 0; g < n && !known; ++g) {
              const uint64_t* rec = &q->records[g * (nwords + 1)];
              known = rec[1] == ident && rec[2] == alt;
            }
            if (known) {
              second = alt;
              value = ~0ull;
            }
          }
        } else {
          if (flags & 2u) ident = ~0ull;
          if (flags & 4u) value = ~0ull;
          second = flags & 1u;  // NULL key
        }
        sets[{ident, second}].push_back(value);
      }
      for (auto& kv : sets) std::sort(kv.second.begin(), kv.second.end());
    }
  }
  q->stats.num_groups = total_groups;
  q->emit_pos = 0;
  q->executed = true;
  q->fetched = true;
  return Status();
}

// ---------------------------------------------------------------------------
// result emission: GroupByExpression::nextBatch (groupby.cc:187-220)
// ---------------------------------------------------------------------------
// EVQL_FLOAT_SUM_EXACT: (sum of high parts) * 2^31 + (sum of low parts) multiples of
// 2^e, rounded to nearest once
static double exact_sum_value(uint64_t hi, uint64_t lo, int e) {
  const __int128 total = (__int128(int64_t(hi)) << 31) + __int128(int64_t(lo));
  return std::ldexp(double(total), e);  // (int128 -> double rounds to nearest even)
}

static Value agg_value(const evql_query* q, const AggPlan& a, const uint64_t* st) {
  Value v;
  v.tag = 0;
  const uint64_t w0 = st[a.first_word];
  switch (a.fn) {
    case EVQL_AGG_COUNT:
    case EVQL_AGG_SUM_UINT64:
    case EVQL_AGG_COUNT_DISTINCT_UINT64:
      v.type = EVQL_T_UINT64;
      v.bits = w0;
      break;
    case EVQL_AGG_SUM_INT64:
      v.type = EVQL_T_INT64;
      v.bits = w0;
      break;
    case EVQL_AGG_SUM_FLOAT64:
      v.type = EVQL_T_FLOAT64;
      v.bits = w0;
      if (a.exact_index >= 0) {
        const double d = exact_sum_value(w0, st[a.first_word + 1], q->fsum_exp[a.exact_index]);
        memcpy(&v.bits, &d, 8);
      }
      break;
    case EVQL_AGG_MIN_UINT64:
    case EVQL_AGG_MAX_UINT64:
    case EVQL_AGG_MIN_INT64:
    case EVQL_AGG_MAX_INT64:
    case EVQL_AGG_MIN_FLOAT64:
    case EVQL_AGG_MAX_FLOAT64: {
      v.type = (a.fn <= EVQL_AGG_MAX_UINT64) ? EVQL_T_UINT64
               : (a.fn <= EVQL_AGG_MAX_INT64 ? EVQL_T_INT64 : EVQL_T_FLOAT64);
      const uint64_t cnt = st[a.first_word + 1];
      if (cnt == 0) {
        v.bits = 0;
        v.tag = EVQL_STAG_NULL;
      } else {
        v.bits = w0;
      }
      break;
    }
    default: {  // mean
      v.type = EVQL_T_FLOAT64;
      const uint64_t cnt = st[a.first_word + 1];
      if (cnt == 0) {
        v.bits = 0;
        v.tag = EVQL_STAG_NULL;
      } else {
        double sum, m;
        memcpy(&sum, &w0, 8);
        m = sum / double(cnt);
        memcpy(&v.bits, &m, 8);
      }
    }
  }
  (void) q;
  return v;
}

static void put_varuint(std::vector<uint8_t>* b, uint64_t v) {
  do {
    uint8_t x = v & 0x7f;
    v >>= 7;
    if (v) x |= 0x80;
    b->push_back(x);
  } while (v);
}

// instance_savestate of each aggregate: count / sum = LEB128 varuint
// (aggregate.cc:55-57,171-173,207-209); build-supplied: sum_float64 = 8 raw
// bytes, min/max/mean = varuint(non-null count) + 8 raw bytes
static void save_state(const evql_query* q, const AggPlan& a, const uint64_t* st,
                       std::vector<uint8_t>* out, const uint64_t* rec = nullptr) {
  const uint64_t w0 = st[a.first_word];
  if (a.fn == EVQL_AGG_COUNT_DISTINCT_UINT64) {
    // varuint size, then the values ascending (std::set order, aggregate.cc:111-117)
    static const std::vector<uint64_t> none;
    const std::vector<uint64_t>* vals = &none;
    if (rec && a.distinct_index >= 0 && size_t(a.distinct_index) < q->distinct_values.size()) {
      const uint64_t kind = rec[0];
      const uint64_t ident = kind == 1 ? ~0ull : (kind == 2 ? 0 : rec[1]);
      const uint64_t second = q->kp.key_mode == KEY_HASHED ? rec[2] : (kind == 2 ? 1 : 0);
      const auto& sets = q->distinct_values[a.distinct_index];
      auto hit = sets.find({q->kp.key_mode == KEY_NONE ? 0 : ident, second});
      if (hit != sets.end()) vals = &hit->second;
    }
    put_varuint(out, vals->size());
    for (uint64_t v : *vals) put_varuint(out, v);
    return;
  }
  const uint8_t* p = reinterpret_cast<const uint8_t*>(&w0);
  switch (a.fn) {
    case EVQL_AGG_COUNT:
    case EVQL_AGG_SUM_UINT64:
    case EVQL_AGG_SUM_INT64:
      put_varuint(out, w0);
      return;
    case EVQL_AGG_SUM_FLOAT64:
      if (a.exact_index >= 0) {  // the wire carries the rounded double
        const double d = exact_sum_value(w0, st[a.first_word + 1], q->fsum_exp[a.exact_index]);
        const uint8_t* pd = reinterpret_cast<const uint8_t*>(&d);
        out->insert(out->end(), pd, pd + 8);
        return;
      }
      out->insert(out->end(), p, p + 8);
      return;
    default: {
      // min / max / mean.  A group without a non-NULL value: the state word still holds the
      // operation's identity (min: all ones) -- on the wire an untouched state is 0
      const uint64_t cnt = st[a.first_word + 1];
      put_varuint(out, cnt);
      static const uint8_t zero[8] = {0};
      if (cnt == 0) {
        out->insert(out->end(), zero, zero + 8);
      } else {
        out->insert(out->end(), p, p + 8);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// ORDER BY .. LIMIT fused above the GROUP BY (orderby.cc:60-160, limit.cc:52-125)
// ---------------------------------------------------------------------------
Status query_set_order(evql_query* q, const evql_sort_spec_t* specs, uint32_t n, int64_t limit,
                       uint64_t offset) {
  if (q->group_mode == EVQL_MODE_PARTIAL) {
    return Status::error(EVQL_EARG, "ORDER BY / LIMIT above a partial aggregate");
  }
  if (n == 0 && limit < 0) {
    return Status::error(EVQL_EARG, "can't execute ORDER BY: no sort specs");  // orderby.cc:53
  }
  std::vector<LoweredProgram> order(n);
  std::vector<bool> desc(n);
  for (uint32_t i = 0; i < n; ++i) {
    bool unsup = false;
    std::string e = lower_program(specs[i].expr, &order[i], &unsup);
    if (!e.empty()) return Status::error(unsup ? EVQL_ENOTSUP : EVQL_EARG, e);
    if (order[i].is_aggregate) return Status::error(EVQL_EARG, "aggregate in ORDER BY");
    std::vector<uint32_t> ins;
    expr_inputs(order[i].call, &ins);
    for (uint32_t in : ins) {
      if (in >= q->select.size()) return Status::error(EVQL_EARG, "invalid input index");
    }
    switch (order[i].return_type) {  // there is no cmp#int64/bool;bool;
      case EVQL_T_UINT64: case EVQL_T_INT64: case EVQL_T_FLOAT64: case EVQL_T_TIMESTAMP64:
      case EVQL_T_STRING:
        break;
      default:
        return Status::error(EVQL_EARG, "no comparator for the sort expression's type");
    }
    desc[i] = specs[i].descending != 0;
  }
  OrderKeyArgs ok{};
  if (n > 0) {
    const KernelPlan& kp = q->rplan();
    const ExprPtr& e0 = order[0].call;
    if (e0->kind != Expr::INPUT) {
      return Status::error(EVQL_ENOTSUP, "first sort expression is not a plain output column");
    }
    const uint32_t si = e0->input;
    const LoweredProgram& sp = q->select[si];
    auto type_code = [](uint32_t t) { return t == EVQL_T_INT64 ? 1u : (t == EVQL_T_FLOAT64 ? 2u : 0u); };
    ok.count_word = -1;
    ok.descending = desc[0];
    if (q->select_passthrough[si] && sp.return_type != EVQL_T_STRING) {
      ok.from_ident = 1;
      ok.word = 1;
      ok.type = type_code(sp.return_type);
    } else if (sp.is_aggregate && sp.call->kind == Expr::AGG_GET) {
      const AggPlan& a = kp.aggs[q->select_agg_index[si]];
      ok.word = uint32_t(1 + kp.state_word_base() + a.first_word);
      switch (a.fn) {
        case EVQL_AGG_COUNT:
        case EVQL_AGG_COUNT_DISTINCT_UINT64:  // (one word: the number of distinct values)
        case EVQL_AGG_SUM_UINT64: ok.type = 0; break;
        case EVQL_AGG_SUM_INT64: ok.type = 1; break;
        case EVQL_AGG_SUM_FLOAT64:
          if (a.exact_index >= 0) {
            return Status::error(EVQL_ENOTSUP, "ORDER BY an exact float sum is not fused");
          }
          ok.type = 2;
          break;
        case EVQL_AGG_MEAN_UINT64:
        case EVQL_AGG_MEAN_INT64:
        case EVQL_AGG_MEAN_FLOAT64:
          ok.type = 2;
          ok.is_mean = 1;
          ok.count_word = int32_t(ok.word + 1);
          break;
        default:  // min / max
          ok.type = type_code(sp.return_type);
          ok.count_word = int32_t(ok.word + 1);
      }
    } else {
      return Status::error(EVQL_ENOTSUP,
                           "first sort expression cannot be read from a group record");
    }
  }
  q->order.swap(order);
  q->order_desc.swap(desc);
  q->has_limit = limit >= 0;
  q->limit = limit >= 0 ? uint64_t(limit) : 0;
  q->offset = offset;
  q->order_key = ok;
  q->fetched = false;
  return Status();
}

// cmp#int64/X;X; (boolean.cc:81-180): payloads only
static int value_cmp(uint32_t type, const Value& a, const Value& b) {
  switch (type) {
    case EVQL_T_INT64: {
      const int64_t l = int64_t(a.bits), r = int64_t(b.bits);
      return l < r ? -1 : (l > r ? 1 : 0);
    }
    case EVQL_T_FLOAT64: {
      double l, r;
      memcpy(&l, &a.bits, 8);
      memcpy(&r, &b.bits, 8);
      return l < r ? -1 : (l > r ? 1 : 0);
    }
    case EVQL_T_STRING: {
      const size_t m = std::min(a.str.size(), b.str.size());
      const int c = m ? strncmp(a.str.data(), b.str.data(), m) : 0;
      if (c != 0) return c < 0 ? -1 : 1;
      return a.str.size() < b.str.size() ? -1 : (a.str.size() > b.str.size() ? 1 : 0);
    }
    default:
      return a.bits < b.bits ? -1 : (a.bits > b.bits ? 1 : 0);
  }
}

static Status final_row_values(evql_query* q, uint64_t g, std::vector<Value>* outs);

// orders the fetched records by every sort spec and applies OFFSET / LIMIT
static Status order_fetched(evql_query* q) {
  const uint64_t n = q->ngroups;
  q->emit_order.clear();
  if (q->order.empty() && !q->has_limit) return Status();
  std::vector<uint64_t> idx(n);
  for (uint64_t i = 0; i < n; ++i) idx[i] = i;
  if (!q->order.empty()) {
    const size_t ns = q->order.size();
    std::vector<Value> keys(n * ns), row;
    for (uint64_t g = 0; g < n; ++g) {
      Status st = final_row_values(q, g, &row);
      if (!st.ok()) return st;
      for (size_t j = 0; j < ns; ++j) {
        std::string e = eval_expr(q->order[j].call, row, nullptr, &keys[g * ns + j]);
        if (!e.empty()) return Status::error(EVQL_ERUNTIME, e);
      }
    }
    std::stable_sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) {
      for (size_t j = 0; j < ns; ++j) {
        const int c = value_cmp(q->order[j].return_type, keys[a * ns + j], keys[b * ns + j]);
        if (c != 0) return q->order_desc[j] ? c > 0 : c < 0;
      }
      return false;
    });
  }
  uint64_t lo = 0, hi = n;
  if (q->has_limit) {
    lo = std::min(q->offset, n);
    hi = q->limit == 0 ? lo : std::min(n, q->offset + q->limit);
  }
  q->emit_order.assign(idx.begin() + lo, idx.begin() + hi);
  return Status();
}

Status query_next_batch(evql_query* q, size_t max_rows, evql_column_buf_t* cols, size_t* nrows) {
  if (!q->executed) return Status::error(EVQL_EARG, "execute() was not called");
  if (!q->fetched) {
    Status st = fetch_results(q);
    if (!st.ok()) return st;
    st = order_fetched(q);
    if (!st.ok()) return st;
  }
  if (q->demit.active) {
    // slices of the columns packed on the device (emit_on_device)
    const evql_query::DeviceEmit& de = q->demit;
    const uint64_t left = q->ngroups - q->emit_pos;
    const uint64_t m = std::min<uint64_t>(left, max_rows);
    for (size_t i = 0; i < de.col.size(); ++i) {
      if (de.elem[i]) {
        cols[i].data = de.col[i] + q->emit_pos * de.elem[i];
        cols[i].size = size_t(m) * de.elem[i];
      } else {
        const uint64_t b0 = de.off[i][q->emit_pos], b1 = de.off[i][q->emit_pos + m];
        cols[i].data = de.col[i] + b0;
        cols[i].size = size_t(b1 - b0);
      }
    }
    q->emit_pos += m;
    *nrows = size_t(m);
    return Status();
  }
  const KernelPlan& kp = q->rplan();
  const size_t nsel = q->select.size();
  const bool partial = q->group_mode == EVQL_MODE_PARTIAL;
  q->out_cols.assign(partial ? 2 : nsel, std::vector<uint8_t>());
  const size_t rw = q->rec_stride;
  const uint32_t nc = uint32_t(kp.cols.size());
  size_t emitted = 0;
  std::vector<Value> scan_vals(nc), sel_inputs(q->scan_select.size());
  const bool reordered = !q->order.empty() || q->has_limit;
  const uint64_t emit_total = reordered ? q->emit_order.size() : q->ngroups;
  while (q->emit_pos < emit_total && emitted < max_rows) {
    const uint64_t g = reordered ? q->emit_order[q->emit_pos] : q->emit_pos;
    const uint64_t* rec = &q->records[g * rw];
    const uint64_t kind = rec[0], ident = rec[1];
    const uint64_t* st = rec + 1 + kp.state_word_base();
    bool have_inputs = false;
    if (kp.need_first_row) {
      for (uint32_t c = 0; c < nc; ++c) {
        const ColAccess& ca = kp.cols[c];
        Value v;
        v.type = ca.stype;
        v.tag = q->first_tags[uint64_t(c) * q->ngroups + g];
        const uint64_t raw = q->first_vals[uint64_t(c) * q->ngroups + g];
        if (ca.string_hash) {
          if (!v.tag) {
            const uint64_t off = q->first_str_off[uint64_t(c) * q->ngroups + g];
            v.str.assign(reinterpret_cast<const char*>(q->first_str_heap.data()) + off,
                         size_t(raw >> 40));
            // A cell of the Dremel scan is a boxed SValue (CSTableScan.cc:300-330): a string
            // of 11 bytes is exactly the 16 bytes of the inline buffer, whose last byte --
            // the value's tag -- also holds STAG_INLINE (svalue.cc:346-368).  X_INPUT copies
            // the bytes as they lie, so the tag 0x80 reaches the group key's SHA1
            // (PartialGroupBy keys) and the output vectors.  Found by the round-3 soak.
            if (q->nested && v.str.size() == 11) v.tag = 0x80;
          }
        } else if (ca.stype == EVQL_T_FLOAT64 && ca.from_uint_to_float) {
          double d = double(raw);
          memcpy(&v.bits, &d, 8);
        } else if (ca.stype == EVQL_T_BOOL) {
          v.bits = raw != 0;
        } else {
          v.bits = raw;
        }
        scan_vals[c] = v;
      }
      for (size_t j = 0; j < q->scan_select.size(); ++j) {
        std::string e = eval_expr(q->scan_select[j].call, scan_vals, nullptr, &sel_inputs[j]);
        if (!e.empty()) return Status::error(EVQL_ERUNTIME, e);
      }
      have_inputs = true;
    }
    std::vector<uint8_t> pdata;  // PARTIAL mode: concatenated saved states
    for (size_t i = 0; i < nsel; ++i) {
      const LoweredProgram& lp = q->select[i];
      Value out;
      if (partial && lp.is_aggregate) {
        save_state(q, kp.aggs[q->select_agg_index[i]], st, &pdata, rec);
        continue;
      }
      if (lp.is_aggregate) {
        Value av = agg_value(q, kp.aggs[q->select_agg_index[i]], st);
        std::string e = eval_expr(lp.call, have_inputs ? sel_inputs : std::vector<Value>(), &av, &out);
        if (!e.empty()) return Status::error(EVQL_ERUNTIME, e);
      } else if (q->select_passthrough[i]) {
        out.type = lp.return_type;
        if (kind == 2) {
          out.bits = 0;
          out.tag = EVQL_STAG_NULL;
        } else {
          out.bits = ident;
          out.tag = 0;
        }
      } else {
        std::string e = eval_expr(lp.call, have_inputs ? sel_inputs : std::vector<Value>(), nullptr, &out);
        if (!e.empty()) return Status::error(EVQL_ERUNTIME, e);
      }
      if (partial) {
        // SValue::encode (svalue.cc): u8 type, lenenc(value || tag bytes)
        std::vector<uint8_t> enc;
        append_svector(lp.return_type, out, &enc);
        // A reference quirk kept for byte-identical wire rows: the value is boxed in an
        // SValue whose 16-byte inline buffer ends in its tag byte, where setData also keeps
        // the internal STAG_INLINE flag (svalue.cc:346-368).  A value of exactly 16 bytes --
        // a string of 11 -- therefore leaves with bit 7 set in its tag.
        if (enc.size() == 16) enc.back() |= 0x80;
        pdata.push_back(uint8_t(lp.return_type));
        put_varuint(&pdata, enc.size());
        pdata.insert(pdata.end(), enc.begin(), enc.end());
      } else {
        append_svector(lp.return_type, out, &q->out_cols[i]);
      }
    }
    if (partial) {
      // group key = SHA1 of the tuple bytes, LAST group expression first
      // (groupby.cc:112-135: the VM stack grows downward)
      std::vector<uint8_t> tuple;
      for (size_t gi = q->group.size(); gi-- > 0;) {
        Value gv;
        if (kp.key_mode == KEY_EXACT) {
          gv.type = q->group[gi].return_type;
          gv.bits = kind == 2 ? 0 : ident;
          gv.tag = kind == 2 ? EVQL_STAG_NULL : 0;
        } else {
          std::string e = eval_expr(q->group[gi].call, sel_inputs, nullptr, &gv);
          if (!e.empty()) return Status::error(EVQL_ERUNTIME, e);
        }
        append_svector(q->group[gi].return_type, gv, &tuple);
      }
      Sha1Digest d = sha1(tuple.data(), tuple.size());
      Value kv, dv;
      kv.str.assign(reinterpret_cast<const char*>(d.bytes), 20);
      dv.str.assign(reinterpret_cast<const char*>(pdata.data()), pdata.size());
      append_svector(EVQL_T_STRING, kv, &q->out_cols[0]);
      append_svector(EVQL_T_STRING, dv, &q->out_cols[1]);
    }
    ++q->emit_pos;
    ++emitted;
  }
  for (size_t i = 0; i < q->out_cols.size(); ++i) {
    cols[i].data = q->out_cols[i].data();
    cols[i].size = q->out_cols[i].size();
  }
  *nrows = emitted;
  return Status();
}

// the select-list values of fetched record g (EVQL_MODE_FINAL), as the emission
// loop above computes them; ORDER BY evaluates its sort expressions over these
static Status final_row_values(evql_query* q, uint64_t g, std::vector<Value>* outs) {
  const KernelPlan& kp = q->rplan();
  const size_t nsel = q->select.size();
  const size_t rw = q->rec_stride;
  const uint32_t nc = uint32_t(kp.cols.size());
  const uint64_t* rec = &q->records[g * rw];
  const uint64_t kind = rec[0], ident = rec[1];
  const uint64_t* st = rec + 1 + kp.state_word_base();
  std::vector<Value> scan_vals(nc), sel_inputs(q->scan_select.size());
  bool have_inputs = false;
  if (kp.need_first_row) {
    for (uint32_t c = 0; c < nc; ++c) {
      const ColAccess& ca = kp.cols[c];
      Value v;
      v.type = ca.stype;
      v.tag = q->first_tags[uint64_t(c) * q->ngroups + g];
      const uint64_t raw = q->first_vals[uint64_t(c) * q->ngroups + g];
      if (ca.string_hash) {
        if (!v.tag) {
          const uint64_t off = q->first_str_off[uint64_t(c) * q->ngroups + g];
          v.str.assign(reinterpret_cast<const char*>(q->first_str_heap.data()) + off,
                       size_t(raw >> 40));
          if (q->nested && v.str.size() == 11) v.tag = 0x80;  // (boxed cell, see query_next_batch)
        }
      } else if (ca.stype == EVQL_T_FLOAT64 && ca.from_uint_to_float) {
        double d = double(raw);
        memcpy(&v.bits, &d, 8);
      } else if (ca.stype == EVQL_T_BOOL) {
        v.bits = raw != 0;
      } else {
        v.bits = raw;
      }
      scan_vals[c] = v;
    }
    for (size_t j = 0; j < q->scan_select.size(); ++j) {
      std::string e = eval_expr(q->scan_select[j].call, scan_vals, nullptr, &sel_inputs[j]);
      if (!e.empty()) return Status::error(EVQL_ERUNTIME, e);
    }
    have_inputs = true;
  }
  const std::vector<Value> none;
  outs->assign(nsel, Value());
  for (size_t i = 0; i < nsel; ++i) {
    const LoweredProgram& lp = q->select[i];
    Value& out = (*outs)[i];
    if (lp.is_aggregate) {
      Value av = agg_value(q, kp.aggs[q->select_agg_index[i]], st);
      std::string e = eval_expr(lp.call, have_inputs ? sel_inputs : none, &av, &out);
      if (!e.empty()) return Status::error(EVQL_ERUNTIME, e);
    } else if (q->select_passthrough[i]) {
      out.type = lp.return_type;
      out.bits = kind == 2 ? 0 : ident;
      out.tag = kind == 2 ? uint8_t(EVQL_STAG_NULL) : uint8_t(0);
    } else {
      std::string e = eval_expr(lp.call, have_inputs ? sel_inputs : none, nullptr, &out);
      if (!e.empty()) return Status::error(EVQL_ERUNTIME, e);
    }
    out.type = lp.return_type;
  }
  return Status();
}

}  // namespace evql
