// lsm.cc -- row filters for a chain of LSM cstable files, built on the device.
//
// Reference: PartitionCursor::openNextTable, server/sql/partition_cursor.cc:83-226.
// The cursor visits head arena, compacting arena and the LSM files newest first;
// for every row it reads `__lsm_id` (a 20-byte SHA1), `__lsm_is_update` and, when
// the file has a skiplist, `__lsm_skip` (arenas: the in-memory skiplist), and
// builds a vector<bool> that FastCSTableScan::setFilter / CSTableScan consume
// (CSTableScan.cc:826-833, 1006-1009):
//
//     if (skip || id_set.count(id))  filter[i] = false;
//     else { if (is_update) id_set.insert(id);  filter[i] = true; }
//
// `id_set` carries over from file to file, so the filter of a row depends on
// every row in front of it.  Per id this is: keep the non-skipped rows up to and
// including the first non-skipped update in scan order, drop everything after.
// That form is order-free: pass 1 records the minimum scan position of a
// non-skipped update per id in an HBM hash table (k_lsm_insert), pass 2 compares
// each row's position with it and packs the bits (k_lsm_filter).
#include <cstring>
#include "aot_kernels.h"
#include "runtime.h"

using namespace evql;

#define LSM_HIP(expr)                                                                    \
  do {                                                                                   \
    hipError_t e__ = (expr);                                                             \
    if (e__ != hipSuccess) {                                                             \
      return Status::error(EVQL_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    }                                                                                    \
  } while (0)

struct evql_lsm_chain {
  evql_ctx* ctx = nullptr;
  struct Entry {
    evql_table* table = nullptr;
    bool has_skip_column = false;
    std::vector<uint8_t> arena_skip;  // one byte per row; empty = none
    std::vector<uint8_t> bits;        // result, ceil(rows / 64) * 8 bytes
    uint64_t kept = 0;
  };
  std::vector<Entry> entries;
  bool built = false;
};

namespace {

Status build(evql_lsm_chain* ch) {
  hipStream_t s = ch->ctx->stream;
  uint64_t total = 0;
  for (const auto& e : ch->entries) total += e.table->layout.num_rows;
  uint64_t cap = 1024;
  while (cap < 2 * total) cap <<= 1;
  uint64_t* d_tab = nullptr;
  uint64_t* d_counters = nullptr;
  LSM_HIP(hipMalloc(reinterpret_cast<void**>(&d_tab), cap * 4 * 8));
  LSM_HIP(hipMalloc(reinterpret_cast<void**>(&d_counters), 16));
  LSM_HIP(hipMemsetAsync(d_tab, 0xff, cap * 4 * 8, s));
  LSM_HIP(hipMemsetAsync(d_counters, 0, 16, s));

  std::vector<LsmArgs> args(ch->entries.size());
  std::vector<uint8_t*> d_skips(ch->entries.size(), nullptr);
  std::vector<uint64_t*> d_bits(ch->entries.size(), nullptr);
  Status st;
  uint64_t pos0 = 0;
  for (size_t i = 0; i < ch->entries.size() && st.ok(); ++i) {
    evql_lsm_chain::Entry& e = ch->entries[i];
    LsmArgs& a = args[i];
    memset(&a, 0, sizeof(a));
    a.image = e.table->d_image;
    a.nrows = e.table->layout.num_rows;
    a.pos0 = pos0;
    pos0 += a.nrows;
    a.tab = d_tab;
    a.cap = cap;
    a.counters = d_counters;
    const uint64_t* strpos = nullptr;
    st = table_rt_column(e.table, "__lsm_id", &a.id, &strpos);
    if (!st.ok()) break;
    if (!strpos) {
      st = Status::error(EVQL_EARG, "__lsm_id is not a string column");
      break;
    }
    a.id_pos = strpos;
    st = table_rt_column(e.table, "__lsm_is_update", &a.is_update, nullptr);
    if (!st.ok()) break;
    if (e.has_skip_column) {
      st = table_rt_column(e.table, "__lsm_skip", &a.skip, nullptr);
      if (!st.ok()) break;
      a.has_skip = 1;
    }
    if (!e.arena_skip.empty()) {
      hipError_t he = hipMalloc(reinterpret_cast<void**>(&d_skips[i]), e.arena_skip.size());
      if (he == hipSuccess) {
        he = hipMemcpyAsync(d_skips[i], e.arena_skip.data(), e.arena_skip.size(),
                            hipMemcpyHostToDevice, s);
      }
      if (he != hipSuccess) {
        st = Status::error(EVQL_EDEVICE, hipGetErrorString(he));
        break;
      }
      a.arena_skip = d_skips[i];
    }
    const uint64_t nwords = (a.nrows + 63) / 64;
    hipError_t he = hipMalloc(reinterpret_cast<void**>(&d_bits[i]), (nwords ? nwords : 1) * 8);
    if (he != hipSuccess) {
      st = Status::error(EVQL_EDEVICE, hipGetErrorString(he));
      break;
    }
    a.bits = d_bits[i];
  }
  if (st.ok()) {
    for (size_t i = 0; i < args.size() && st.ok(); ++i) {
      hipError_t he = launch_lsm_insert(args[i], s);
      if (he != hipSuccess) st = Status::error(EVQL_EDEVICE, hipGetErrorString(he));
    }
  }
  uint64_t prev_kept = 0;
  for (size_t i = 0; i < args.size() && st.ok(); ++i) {
    hipError_t he = launch_lsm_filter(args[i], s);
    if (he != hipSuccess) {
      st = Status::error(EVQL_EDEVICE, hipGetErrorString(he));
      break;
    }
    evql_lsm_chain::Entry& e = ch->entries[i];
    const uint64_t nwords = (args[i].nrows + 63) / 64;
    e.bits.assign(nwords * 8, 0);
    uint64_t counters[2] = {0, 0};
    if (nwords) he = hipMemcpyAsync(e.bits.data(), d_bits[i], nwords * 8, hipMemcpyDeviceToHost, s);
    if (he == hipSuccess) he = hipMemcpyAsync(counters, d_counters, 16, hipMemcpyDeviceToHost, s);
    if (he == hipSuccess) he = hipStreamSynchronize(s);
    if (he != hipSuccess) {
      st = Status::error(EVQL_EDEVICE, hipGetErrorString(he));
      break;
    }
    if (counters[1]) {
      st = Status::error(EVQL_ERUNTIME, "invalid SHA1Hash");  // util/SHA1.cc:79-85
      break;
    }
    e.kept = counters[0] - prev_kept;
    prev_kept = counters[0];
  }
  hipStreamSynchronize(s);
  for (auto* p : d_skips) {
    if (p) hipFree(p);
  }
  for (auto* p : d_bits) {
    if (p) hipFree(p);
  }
  hipFree(d_tab);
  hipFree(d_counters);
  return st;
}

}  // namespace

extern "C" {

int evql_lsm_chain_create(evql_ctx_t* ctx, evql_lsm_chain_t** out) {
  if (!ctx || !out) return fail(EVQL_EARG, "null argument");
  evql_lsm_chain* ch = new evql_lsm_chain();
  ch->ctx = ctx;
  *out = ch;
  return EVQL_OK;
}

void evql_lsm_chain_destroy(evql_lsm_chain_t* ch) { delete ch; }

int evql_lsm_chain_add(evql_lsm_chain_t* ch, evql_table_t* table, int has_skip_column,
                       const uint8_t* arena_skiplist, uint64_t arena_skiplist_len) {
  if (!ch || !table) return fail(EVQL_EARG, "null argument");
  if (table->ctx != ch->ctx) return fail(EVQL_EARG, "table belongs to another context");
  if (arena_skiplist && arena_skiplist_len != table->layout.num_rows) {
    return fail(EVQL_EARG, "arena skiplist length differs from the table's row count");
  }
  evql_lsm_chain::Entry e;
  e.table = table;
  e.has_skip_column = has_skip_column != 0;
  if (arena_skiplist) e.arena_skip.assign(arena_skiplist, arena_skiplist + arena_skiplist_len);
  ch->entries.push_back(std::move(e));
  ch->built = false;
  return EVQL_OK;
}

int evql_lsm_chain_build(evql_lsm_chain_t* ch) {
  if (!ch) return fail(EVQL_EARG, "null argument");
  Status st = build(ch);
  if (!st.ok()) return fail(st.code, st.msg);
  ch->built = true;
  return EVQL_OK;
}

int evql_lsm_chain_filter(const evql_lsm_chain_t* ch, int idx, const uint8_t** bits,
                          uint64_t* nrows, uint64_t* rows_kept) {
  if (!ch || !ch->built) return fail(EVQL_EARG, "chain was not built");
  if (idx < 0 || size_t(idx) >= ch->entries.size()) return fail(EVQL_EARG, "bad table index");
  const evql_lsm_chain::Entry& e = ch->entries[idx];
  if (bits) *bits = e.bits.data();
  if (nrows) *nrows = e.table->layout.num_rows;
  if (rows_kept) *rows_kept = e.kept;
  return EVQL_OK;
}

}  // extern "C"
