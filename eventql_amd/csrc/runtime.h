// runtime.h -- device context, HBM-resident tables and query objects behind the
// C ABI (include/evql_gpu.h).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "aot_kernels.h"
#include "codegen.h"
#include "cstable_format.h"
#include "plan_ir.h"

namespace evql {

// host mirror of the device-side EvqlArgs / EvqlColArg (evql_device.h)
struct HostColArg {
  const uint64_t* pages;
  const uint64_t* soa;
  const uint8_t* tags;
  uint64_t npages;
  const uint64_t* strpos;
  const uint8_t* base;
};
static const uint32_t EVQL_MAX_COLS_HOST = 16;
struct HostArgs {
  const uint8_t* image;
  uint64_t row_begin, row_end, ntiles, tile0;
  const uint8_t* row_filter;
  uint64_t row_filter_len;
  uint64_t* gtab;
  uint64_t gcap;
  uint32_t* status;
  uint64_t* counters;
  HostColArg col[16];
  uint64_t* pairset[4];
  uint64_t pairset_cap[4];
  double fscale[4];
  double fbound[4];
};

struct Status {
  int code = EVQL_OK;
  std::string msg;
  bool ok() const { return code == EVQL_OK; }
  static Status error(int c, const std::string& m) {
    Status s;
    s.code = c;
    s.msg = m;
    return s;
  }
};

struct Module {
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  // partitioned high-cardinality path (present only when the plan asks for it)
  hipFunction_t fn_count = nullptr, fn_scatter = nullptr, fn_aggregate = nullptr;
  hipFunction_t fn_refine = nullptr;  // second scatter level (part_bits > 8)
  hipFunction_t fn_where = nullptr;   // evql_where_rows (nested scans, mixed-depth WHERE)
  size_t code_size = 0;
};

// host mirror of the generated EvqlPartArgs
struct HostPartArgs {
  uint64_t tiles_per_wg;
  uint32_t* counts;
  const uint64_t* bucket_start;
  uint64_t* tuples;
  uint64_t nwg;
  uint64_t* tuples_tmp;
  uint32_t* cursors;
  uint64_t* dense;
  uint64_t dense_cap;
  uint64_t coarse_cap;  // tuples per coarse bucket of tuples_tmp (KernelPlan::part_fused)
};
struct HostArgsWithPart {
  HostArgs a;
  HostPartArgs p;
};

}  // namespace evql

struct evql_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int num_cus = 256;
  std::map<std::string, evql::Module> modules;  // by source fingerprint
};

namespace evql {
// owning device pointer for temporaries: freed on every exit path (HIP_TRY returns
// early on errors)
template <typename T>
struct DevBuf {
  T* p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { reset(); }
  hipError_t alloc(size_t bytes) {
    reset();
    return hipMalloc(reinterpret_cast<void**>(&p), bytes ? bytes : 1);
  }
  void reset() {
    if (p) hipFree(p);
    p = nullptr;
  }
  T* release() {
    T* r = p;
    p = nullptr;
    return r;
  }
  operator T*() const { return p; }
};
}  // namespace evql

// a decoded-to-SoA column cached on the table (owns its device arrays)
struct MaterializedColumn {
  uint64_t* d_values = nullptr;
  uint8_t* d_tags = nullptr;
  // UINT64_LEB128 re-encoded as bit-packed pages of width 8 / 16 / 32 (the narrowest
  // that holds the column's maximum; widths that divide 32 decode with one shift):
  // the fused kernel then reads 1 - 4 bytes per value instead of an 8-byte SoA word
  uint8_t* d_packed = nullptr;
  uint64_t* d_packed_pages = nullptr;  // page offsets into d_packed
  uint32_t packed_bits = 0;
  bool string_hash = false;
  // strings: (len << 40) | byte position of the value in the column's page stream,
  // per row (bytewise compares in the fused kernel, result emission)
  uint64_t* d_strpos = nullptr;

  MaterializedColumn() = default;
  MaterializedColumn(const MaterializedColumn&) = delete;
  MaterializedColumn& operator=(const MaterializedColumn&) = delete;
  MaterializedColumn(MaterializedColumn&& o) noexcept { *this = std::move(o); }
  MaterializedColumn& operator=(MaterializedColumn&& o) noexcept {
    std::swap(d_values, o.d_values);
    std::swap(d_tags, o.d_tags);
    std::swap(d_strpos, o.d_strpos);
    std::swap(string_hash, o.string_hash);
    std::swap(d_packed, o.d_packed);
    std::swap(d_packed_pages, o.d_packed_pages);
    std::swap(packed_bits, o.packed_bits);
    return *this;
  }
  ~MaterializedColumn() {
    if (d_values) hipFree(d_values);
    if (d_tags) hipFree(d_tags);
    if (d_strpos) hipFree(d_strpos);
    if (d_packed) hipFree(d_packed);
    if (d_packed_pages) hipFree(d_packed_pages);
  }
};

// Dictionary of a STRING column: one dense 32-bit code per distinct value, EXACT (equal
// codes <=> equal bytes, verified when it is built, string_dict.cc).  A GROUP BY over the
// column then runs on the codes -- a 4-byte exact key instead of 64-bit hashes plus a row
// index -- and its group records are translated back (code -> hashed identity + first
// row) only where they leave the operator.
struct StringDict {
  bool tried = false;
  bool usable = false;
  uint64_t n_codes = 0;
  uint32_t* d_codes = nullptr;       // per row, padded like the other column buffers
  uint64_t* d_code_pages = nullptr;  // "page" offsets into d_codes (131,072 codes each)
  uint64_t* d_entries = nullptr;     // [n_codes][3]: string hash, first row, bit0 NULL
  std::string why;                   // when not usable
  StringDict() = default;
  StringDict(const StringDict&) = delete;
  StringDict& operator=(const StringDict&) = delete;
  ~StringDict() {
    if (d_codes) hipFree(d_codes);
    if (d_code_pages) hipFree(d_code_pages);
    if (d_entries) hipFree(d_entries);
  }
};

struct evql_table {
  evql_ctx* ctx = nullptr;
  evql::TableLayout layout;
  uint8_t* d_image = nullptr;
  uint64_t image_len = 0;
  // device page tables: [column][0 data, 1 rlevel, 2 dlevel]
  std::vector<std::vector<uint64_t*>> d_pages;
  std::vector<uint64_t> payload_bytes;
  std::map<std::string, MaterializedColumn> materialized;
  std::map<std::string, StringDict> dicts;  // by column name
  // maximum |value| per column ("name#f" float view, "name#u" integer view): bounds of
  // exact float sums (EVQL_FLOAT_SUM_EXACT)
  std::map<std::string, double> col_absmax;
  // nested scans: column flattened to one value per output row of the scans whose
  // deepest repeated column is `leaf` -- key (column, leaf) layout indices.  Like
  // `materialized`, decoded once per table and shared by every operator.
  struct NestedFlat {
    uint64_t* d_values = nullptr;  // per row: value bits; strings: (len << 40 | position)
    uint64_t nflat = 0;
    uint64_t* d_hash = nullptr;    // strings: 64-bit hash of the row's bytes
    // the values once more as bit-packed pages of 8 / 16 / 32 bits (when their maximum
    // fits): what the fused kernel streams instead of the 8-byte words
    uint8_t* d_packed = nullptr;
    uint64_t* d_packed_pages = nullptr;
    uint32_t packed_bits = 0;
    bool pack_tried = false;
  };
  std::map<std::pair<int, int>, NestedFlat> nested_cache;
  // record scans (WITHIN RECORD): the leaf's decoded repetition levels (one byte
  // per slot) and the scanned per-tile counts of its level-0 slots (= records
  // started before the tile), keyed by the leaf's layout index
  struct LeafLevels {
    uint8_t* levels = nullptr;
    uint64_t* rec_offsets = nullptr;
  };
  std::map<int, LeafLevels> leaf_cache;
  ~evql_table();
};

struct evql_query {
  evql_ctx* ctx = nullptr;
  evql_table* table = nullptr;
  // plan
  std::vector<evql::LoweredProgram> scan_select, group, select;
  evql::LoweredProgram where;
  bool has_where = false;
  evql::KernelPlan kp;
  // Dictionary-coded string key (StringDict): `kp` is then the plan the kernels run --
  // KEY_EXACT over the code column -- and `rkp` the plan its group RECORDS follow once
  // they leave the scan (hashed string key + first row: what the plan is without a
  // dictionary).  Emission, ORDER BY, the exchange and chain merges read rplan().
  bool dict_key = false;
  int dict_candidate = -1;  // scan column a dictionary could code (planner), or -1
  evql::KernelPlan rkp;
  const evql::KernelPlan& rplan() const { return dict_key ? rkp : kp; }
  uint64_t* d_conv = nullptr;  // the groups as records of rplan()'s layout
  uint64_t conv_cap = 0;       // records
  bool conv_valid = false;
  std::vector<int> select_agg_index;  // select expr -> index into kp.aggs or -1
  std::vector<bool> select_passthrough;
  uint32_t group_mode = EVQL_MODE_FINAL;
  uint64_t groups_hint = 0;
  // EVQL_FLOAT_SUM_EXACT: quantum exponent / bound per exact sum (index AggPlan::exact_index)
  uint32_t float_sum_mode = 0;
  double float_sum_bound = 0;
  int fsum_exp[4] = {0, 0, 0, 0};   // quantum = 2^fsum_exp
  double fsum_bound[4] = {0, 0, 0, 0};
  uint64_t row_begin = 0, row_end = 0;
  std::vector<uint8_t> row_filter_host;
  uint64_t row_filter_len = 0;
  uint8_t* d_row_filter = nullptr;
  bool row_filter_owned = true;  // false: the bits belong to an evql_lsm_chain
  // evql_query_create_chain: this query scans the first table of a partition's chain;
  // `chain` holds the queries of the tables behind it (owned).  After execute their
  // groups are merged in chain order into d_mtab (exchange.cc chain_merge).
  std::vector<evql_query*> chain;
  // partitioned path buffers
  uint32_t* d_part_counts = nullptr;
  uint64_t* d_bucket_start = nullptr;
  uint64_t* d_tuples = nullptr;
  uint64_t* d_tuples_tmp = nullptr;  // coarse-bucket order (two-level scatter)
  uint32_t* d_part_cursors = nullptr;
  uint64_t tuples_cap = 0;  // in tuples
  uint64_t tuples_tmp_cap = 0;  // in tuples
  bool part_fused_off = false;  // a coarse bucket overflowed its slack once: exact offsets from now on
  // partitioned path: the groups of every bucket that fitted its LDS table, as dense
  // records [kind, identity, (identity 2), (first row), states...]; the HBM table
  // then only holds the groups of overflowed buckets
  uint64_t* d_dense = nullptr;
  uint64_t dense_cap = 0;  // records
  uint64_t dense_n = 0;
  // after evql_query_exchange: the merged groups of all ranks (or of this rank's key
  // range) in a table of their own, whose slots also carry the first-row value words
  // of plans that need them (exchange.cc); results are then emitted from here
  bool merged = false;
  uint64_t* d_mtab = nullptr;
  uint64_t mcap = 0;
  uint32_t m_words = 0;
  std::vector<uint8_t> m_heap;  // received string bytes
  // large merges (exchange.cc bucketed_merge) leave dense records [kind, slot words...]
  // instead of a table
  bool merged_dense = false;
  uint64_t* d_mdense = nullptr;
  uint64_t mdense_cap = 0;  // records
  uint64_t mdense_n = 0;
  int n_update_words = 0;   // update words per row (tuple payload)
  // nested (Dremel) scans: flattened per-row SoA columns, one per scan column
  bool nested = false;
  uint64_t nested_rows = 0;
  std::vector<uint64_t*> nested_flat;
  std::vector<uint64_t*> nested_strpos;  // string columns: (len << 40 | position) per row
  // nested scans over narrow bit-packed copies of the flattened columns (ColAccess::packed)
  struct PackedSource {
    const uint8_t* base = nullptr;
    const uint64_t* pages = nullptr;
  };
  std::vector<PackedSource> nested_packed;
  int nested_leaf = -1;  // layout index of the leaf column of the scan
  std::vector<uint64_t*> nested_owned;
  bool nested_where_mixed = false;  // WHERE over columns of different repetition depth
  bool nested_siblings = false;     // columns of sibling repeated groups: zipped (materialize_nested_zip)
  // EVQL_SCAN_NESTED_WITHIN_RECORD (CSTableScan.cc:440-487,
  // AGGREGATE_WITHIN_RECORD_FLAT): the scan select list holds one aggregate per
  // expression, reduced to one value per record; those per-record arrays are
  // the `kp.cols` the operators above the scan read.
  struct WithinAgg {
    bool is_count = false;
    int col = -1;        // index into wr_cols, or -1: literal / no argument
    uint64_t lit = 0;
    uint32_t level = 0;  // select_list_[i].rep_level
  };
  bool within_record = false;
  double within_record_ms = 0;  // device time of k_within_record (part of this operator's work)
  std::vector<evql::ColAccess> wr_cols;  // the columns the record scan reads
  std::vector<WithinAgg> wr_aggs;
  std::string source;
  evql::Module module;
  // execution state
  uint64_t* d_gtab = nullptr;
  uint64_t gcap = 0;
  uint32_t* d_status = nullptr;
  uint64_t* d_counters = nullptr;
  uint64_t* d_small_rec = nullptr;  // 1 MiB: dense records of small results
  // count_distinct pair sets (3 word planes of pairset_cap slots each)
  uint64_t* d_pairset[4] = {nullptr, nullptr, nullptr, nullptr};
  uint64_t pairset_cap = 0;
  // after an exchange / chain merge: the merged pair sets (the union of the ranks' /
  // tables' sets for the groups this query now holds); PARTIAL rows read their values here
  uint64_t* d_mset[4] = {nullptr, nullptr, nullptr, nullptr};
  uint64_t mset_cap[4] = {0, 0, 0, 0};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool probed = false;    // cardinality probe done (plans without groups_hint)
  bool keep_table = false;  // launch without emptying the group table / counters (the
                            // cardinality probe aggregates several row ranges into one)
  // heartbeat of the running evql_query_execute: also beaten between the launches of the
  // cardinality probe and between the re-runs after a full table
  int (*hb)(void*) = nullptr;
  void* hb_user = nullptr;
  bool hb_abort = false;
  bool launched = false;
  bool executed = false;
  bool fetched = false;  // groups copied to the host (lazy, on first nextBatch)
  int grid = 0;
  // results
  std::vector<uint64_t> records;  // dense [kind, ident, (first_row), states...]
  size_t rec_stride = 0;          // words per fetched record (>= words_per_slot + 1)
  uint64_t ngroups = 0;
  std::vector<uint64_t> first_vals;  // [col][group]
  std::vector<uint8_t> first_tags;
  std::vector<uint8_t> first_str_heap;  // bytes of the first-row strings
  std::vector<uint64_t> first_str_off;  // [col][group] offset into the heap
  uint64_t emit_pos = 0;
  std::vector<std::vector<uint8_t>> out_cols;
  // large results packed on the device (runtime.cc emit_on_device): every output column
  // of ALL groups as SVector bytes in pinned host memory, next_batch hands out slices
  struct DeviceEmit {
    bool active = false;
    std::vector<uint8_t*> col;      // pinned (hipHostMalloc), per select expression
    std::vector<size_t> col_cap;
    std::vector<uint32_t> elem;     // bytes per element; 0 = STRING
    std::vector<uint64_t*> off;     // strings: pinned [n + 1] byte offsets
    std::vector<size_t> off_cap;
  } demit;
  // EVQL_MODE_PARTIAL with count_distinct: the distinct values of every group, per
  // aggregate, read back from the HBM pair set; key = (identity, identity 2 | NULL flag)
  std::vector<std::map<std::pair<uint64_t, uint64_t>, std::vector<uint64_t>>> distinct_values;
  // ORDER BY .. LIMIT fused above the GROUP BY (evql_query_set_order)
  std::vector<evql::LoweredProgram> order;  // over the select list
  std::vector<bool> order_desc;
  bool has_limit = false;
  uint64_t limit = 0, offset = 0;
  evql::OrderKeyArgs order_key{};     // where the device finds sort key 0 in a record
  std::vector<uint64_t> emit_order;  // record indices in output order (when ordered)
  evql_query_stats_t stats{};
  ~evql_query();
};

struct evql_writer {
  std::unique_ptr<evql::TableWriter> w;
  std::vector<evql::ColumnSpec> specs;
};

namespace evql {
void set_last_error(const std::string& m);
int fail(int code, const std::string& m);

Status compile_kernel(evql_ctx* ctx, const std::string& source, Module* out,
                      bool load_module);
Status build_kernel_plan(const TableLayout& layout, const evql_plan_desc_t* plan,
                         evql_query* q, bool* unsupported);
// row-addressable device view of a flat column (direct pages or cached SoA decode)
Status table_rt_column(evql_table* t, const std::string& name, RtColumn* out,
                       const uint64_t** strpos);
// device copies of the page offset lists of `t->layout` -> t->d_pages
Status upload_page_tables(evql_table* t);
// string_dict.cc: the (cached) dictionary of STRING column `li`; built on first use
Status table_string_dict(evql_table* t, int li, StringDict** out);
// the query's groups as dense records of rplan()'s layout: `dense` holds the first `nd`
// groups, the remaining ngroups - nd sit in the HBM group table (compact them from there)
struct RecordsView {
  const uint64_t* dense = nullptr;
  uint64_t nd = 0;
};
Status query_records_view(evql_query* q, RecordsView* out);
// device_writer.cc: a cstable v0.2.0 image encoded on the device from SoA columns
struct DeviceColumnIn {
  const uint64_t* values;  // device, num_rows value words
  const uint8_t* nulls;    // device, num_rows bytes (1 = NULL) or nullptr
  const uint8_t* bytes;    // device, string columns: the heap `values` point into
  // repeated / nested columns: one (r, d, value) triple per SLOT
  const uint8_t* rlevels;  // device, num_slots bytes (rlevel_max > 0)
  const uint8_t* dlevels;  // device, num_slots bytes (instead of `nulls`)
  uint64_t num_slots;      // 0 = one slot per row
};
Status table_from_device_columns(evql_ctx* ctx, const std::vector<ColumnSpec>& specs,
                                 const std::vector<DeviceColumnIn>& in, uint64_t num_rows,
                                 int page_order, evql_table** out);
}  // namespace evql
