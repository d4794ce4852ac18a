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
    bool has_skiplist = false;        // LSMTableRef::has_skiplist: a __lsm_skip column
    bool has_updates = true;          // LSMTableRef::has_updates
    std::vector<uint8_t> arena_skip;  // one byte per row; empty = not an arena
    bool is_arena = false;
    bool needs_filter = true;         // partition_cursor.cc:149-155
    uint64_t* d_bits = nullptr;       // device filter words (needs_filter only)
    std::vector<uint8_t> bits;        // host copy, fetched on demand
    bool bits_fetched = false;
    uint64_t kept = 0;
  };
  std::vector<Entry> entries;
  bool built = false;
  ~evql_lsm_chain() { release(); }
  void release() {
    for (auto& e : entries) {
      if (e.d_bits) hipFree(e.d_bits);
      e.d_bits = nullptr;
      e.bits.clear();
      e.bits_fetched = false;
    }
  }
};

namespace {

Status build(evql_lsm_chain* ch) {
  hipStream_t s = ch->ctx->stream;
  ch->release();
  uint64_t total = 0;
  for (const auto& e : ch->entries) total += e.table->layout.num_rows;
  uint64_t cap = 1024;
  while (cap < 2 * total) cap <<= 1;
  DevBuf<uint64_t> d_tab, d_counters;
  LSM_HIP(d_tab.alloc(cap * 4 * 8));
  LSM_HIP(d_counters.alloc(32));
  LSM_HIP(hipMemsetAsync(d_tab, 0xff, cap * 4 * 8, s));
  LSM_HIP(hipMemsetAsync(d_counters, 0, 32, s));

  const size_t n = ch->entries.size();
  std::vector<LsmArgs> args(n);
  std::vector<DevBuf<uint8_t>> d_skips(n);
  // the oldest LSM file of the chain (tblidx == 0): the last entry, unless it is an arena
  const size_t oldest = n ? n - 1 : 0;
  bool id_set_empty = true;
  uint64_t pos0 = 0;
  // pass 1, table by table in scan order: whether a file needs a filter at all depends
  // on the ids remembered so far (partition_cursor.cc:149-155) -- a file that needs
  // none is scanned whole and its updates are NOT remembered
  for (size_t i = 0; i < n; ++i) {
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
    e.needs_filter = true;
    if (!e.is_arena) {
      if (!e.has_skiplist && i == oldest && id_set_empty) e.needs_filter = false;
      if (!e.has_skiplist && !e.has_updates && id_set_empty) e.needs_filter = false;
    }
    e.kept = a.nrows;
    if (!e.needs_filter) continue;
    const uint64_t* strpos = nullptr;
    Status st = table_rt_column(e.table, "__lsm_id", &a.id, &strpos);
    if (!st.ok()) return st;
    if (!strpos) return Status::error(EVQL_EARG, "__lsm_id is not a string column");
    a.id_pos = strpos;
    st = table_rt_column(e.table, "__lsm_is_update", &a.is_update, nullptr);
    if (!st.ok()) return st;
    if (e.has_skiplist && !e.is_arena) {
      st = table_rt_column(e.table, "__lsm_skip", &a.skip, nullptr);
      if (!st.ok()) return st;
      a.has_skip = 1;
    }
    if (e.is_arena) {
      LSM_HIP(d_skips[i].alloc(e.arena_skip.size()));
      LSM_HIP(hipMemcpyAsync(d_skips[i], e.arena_skip.data(), e.arena_skip.size(),
                             hipMemcpyHostToDevice, s));
      a.arena_skip = d_skips[i];
    }
    const uint64_t nwords = (a.nrows + 63) / 64;
    LSM_HIP(hipMalloc(reinterpret_cast<void**>(&e.d_bits), (nwords ? nwords : 1) * 8 + 16));
    a.bits = e.d_bits;
    LSM_HIP(launch_lsm_insert(a, s));
    if (id_set_empty && i + 1 < n) {
      // (only the emptiness of the id set feeds back into the next table's decision)
      uint64_t counters[4] = {0, 0, 0, 0};
      LSM_HIP(hipMemcpyAsync(counters, d_counters, 32, hipMemcpyDeviceToHost, s));
      LSM_HIP(hipStreamSynchronize(s));
      if (counters[1]) return Status::error(EVQL_ERUNTIME, "invalid SHA1Hash");  // util/SHA1.cc:79-85
      id_set_empty = counters[2] == 0;
    }
  }
  // pass 2: one compare per row
  uint64_t prev_kept = 0;
  for (size_t i = 0; i < n; ++i) {
    evql_lsm_chain::Entry& e = ch->entries[i];
    if (!e.needs_filter) continue;
    LSM_HIP(launch_lsm_filter(args[i], s));
    uint64_t counters[4] = {0, 0, 0, 0};
    LSM_HIP(hipMemcpyAsync(counters, d_counters, 32, hipMemcpyDeviceToHost, s));
    LSM_HIP(hipStreamSynchronize(s));
    if (counters[1]) return Status::error(EVQL_ERUNTIME, "invalid SHA1Hash");  // util/SHA1.cc:79-85
    e.kept = counters[0] - prev_kept;
    prev_kept = counters[0];
  }
  LSM_HIP(hipStreamSynchronize(s));
  return Status();
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

int evql_lsm_chain_add(evql_lsm_chain_t* ch, evql_table_t* table, uint32_t flags,
                       const uint8_t* arena_skiplist, uint64_t arena_skiplist_len) {
  if (!ch || !table) return fail(EVQL_EARG, "null argument");
  if (table->ctx != ch->ctx) return fail(EVQL_EARG, "table belongs to another context");
  if (arena_skiplist && arena_skiplist_len != table->layout.num_rows) {
    return fail(EVQL_EARG, "arena skiplist length differs from the table's row count");
  }
  evql_lsm_chain::Entry e;
  e.table = table;
  e.has_skiplist = (flags & EVQL_LSM_HAS_SKIPLIST) != 0;
  e.has_updates = (flags & EVQL_LSM_HAS_UPDATES) != 0;
  e.is_arena = arena_skiplist != nullptr;
  if (arena_skiplist) e.arena_skip.assign(arena_skiplist, arena_skiplist + arena_skiplist_len);
  if (e.is_arena) {
    for (const auto& o : ch->entries) {
      if (!o.is_arena) return fail(EVQL_EARG, "arenas come before the LSM files of a chain");
    }
  }
  ch->entries.push_back(std::move(e));
  ch->built = false;
  return EVQL_OK;
}

int evql_lsm_chain_build(evql_lsm_chain_t* ch) {
  if (!ch) return fail(EVQL_EARG, "null argument");
  if (hipSetDevice(ch->ctx->device) != hipSuccess) return fail(EVQL_EDEVICE, "hipSetDevice failed");
  Status st = build(ch);
  if (!st.ok()) {
    ch->release();
    return fail(st.code, st.msg);
  }
  ch->built = true;
  return EVQL_OK;
}

int evql_lsm_chain_length(const evql_lsm_chain_t* ch) { return ch ? int(ch->entries.size()) : 0; }

int evql_lsm_chain_filter(evql_lsm_chain_t* ch, int idx, const uint8_t** bits, uint64_t* nrows,
                          uint64_t* rows_kept) {
  if (!ch || !ch->built) return fail(EVQL_EARG, "chain was not built");
  if (idx < 0 || size_t(idx) >= ch->entries.size()) return fail(EVQL_EARG, "bad table index");
  evql_lsm_chain::Entry& e = ch->entries[idx];
  if (bits) {
    *bits = nullptr;
    if (e.needs_filter) {
      if (!e.bits_fetched) {
        const uint64_t nwords = (e.table->layout.num_rows + 63) / 64;
        e.bits.assign(nwords * 8, 0);
        if (nwords && hipMemcpy(e.bits.data(), e.d_bits, nwords * 8, hipMemcpyDeviceToHost) != hipSuccess) {
          return fail(EVQL_EDEVICE, "copy of the filter failed");
        }
        e.bits_fetched = true;
      }
      static const uint8_t none = 0;
      *bits = e.bits.empty() ? &none : e.bits.data();
    }
  }
  if (nrows) *nrows = e.table->layout.num_rows;
  if (rows_kept) *rows_kept = e.kept;
  return EVQL_OK;
}

}  // extern "C"

// runtime-internal view of a built chain (capi.cc evql_query_create_chain)
namespace evql {
size_t lsm_chain_parts(const evql_lsm_chain* ch, std::vector<evql_table*>* tables,
                       std::vector<const uint8_t*>* d_filters) {
  tables->clear();
  d_filters->clear();
  if (!ch || !ch->built) return 0;
  for (const auto& e : ch->entries) {
    tables->push_back(e.table);
    d_filters->push_back(e.needs_filter ? reinterpret_cast<const uint8_t*>(e.d_bits) : nullptr);
  }
  return tables->size();
}
evql_ctx* lsm_chain_ctx(const evql_lsm_chain* ch) { return ch ? ch->ctx : nullptr; }
}  // namespace evql
