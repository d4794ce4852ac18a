// string_dict.cc -- per-table dictionaries of STRING columns.
//
// The reference groups by the bytes of the key tuple (SHA1 of them,
// sql/statements/select/groupby.cc:129-135): two rows belong to one group iff their key
// strings are equal.  A GROUP BY over a string column with many distinct values is what
// BASELINE.json's config 4 asks for ("1e7 string-hash groups").  Carrying a string's
// identity through the partitioned path as two 64-bit hash words plus the row index that
// finds the bytes again costs a 32-byte tuple per passing row and pass; a dictionary
// makes the identity a dense 32-bit code -- exact, not a hash -- and the tuple 16 bytes.
//
// Built once per (table, column), cached with the table like the other decoded forms:
//   1. GROUP BY over the column's 64-bit string hashes with count(1) and the first row:
//      the fused / partitioned kernels of an ordinary plan (KEY_EXACT over the hash
//      column), D groups -> record i is code i, its first row the representative;
//   2. an open-addressed table hash -> code (k_dict_insert);
//   3. per row: code by lookup, and the row's bytes compared with the representative's
//      (k_dict_assign).  One mismatch anywhere -- two different strings with one 64-bit
//      hash -- and the dictionary is dropped: the plan then keeps its hashed identity.
#include <cstring>
#include "aot_kernels.h"
#include "runtime.h"

namespace evql {

Status query_prepare(evql_query* q);
Status query_launch(evql_query* q);
Status query_finish(evql_query* q);

#define DICT_HIP(expr)                                                                       \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      return Status::error(EVQL_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    }                                                                                        \
  } while (0)

static Status build_dict(evql_table* t, int li, StringDict* d) {
  evql_ctx* ctx = t->ctx;
  hipStream_t s = ctx->stream;
  const ColumnLayout& cl = t->layout.columns[li];
  const uint64_t n = t->layout.num_rows;
  if (n == 0) {
    d->why = "empty table";
    return Status();
  }
  // ---- 1. the distinct hashes: an ordinary GROUP BY over the hash column ----------------------
  std::unique_ptr<evql_query> dq(new evql_query());
  dq->ctx = ctx;
  dq->table = t;
  KernelPlan& kp = dq->kp;
  ColAccess c;
  c.name = cl.name;
  c.layout_index = li;
  c.stype = EVQL_T_UINT64;  // the materialised 64-bit hash, read as a number
  c.mode = ColAccess::SOA;
  c.has_tags = cl.dlevel_max > 0;
  kp.cols.push_back(c);
  auto in = std::make_shared<Expr>();
  in->kind = Expr::INPUT;
  in->type = EVQL_T_UINT64;
  in->input = 0;
  kp.group.push_back(in);
  kp.key_mode = KEY_EXACT;
  kp.need_first_row = true;
  AggPlan a;
  a.fn = EVQL_AGG_COUNT;
  a.first_word = 0;
  a.nwords = 1;
  kp.aggs.push_back(a);
  kp.states.push_back({0});
  choose_launch_shape(&kp, 0);
  Status st = query_prepare(dq.get());
  if (!st.ok()) return st;
  st = query_launch(dq.get());
  if (st.ok()) st = query_finish(dq.get());
  if (!st.ok()) return st;
  const uint64_t D = dq->ngroups;
  if (D == 0 || D >= 0xfffffffeull) {
    d->why = "no codes / more than 2^32 - 2 distinct values";
    return Status();
  }
  const uint32_t W = uint32_t(kp.words_per_slot());  // identity, first row, count
  DevBuf<uint64_t> d_rec;
  DICT_HIP(d_rec.alloc(D * (W + 1) * 8));
  {
    const uint64_t nd = std::min(dq->dense_n, D);
    if (nd) DICT_HIP(hipMemcpyAsync(d_rec, dq->d_dense, nd * (W + 1) * 8, hipMemcpyDeviceToDevice, s));
    if (D > nd) {
      uint64_t* d_cnt = dq->d_counters + 6;
      DICT_HIP(hipMemsetAsync(d_cnt, 0, 8, s));
      DICT_HIP(launch_table_compact(dq->d_gtab, dq->gcap, dq->gcap + 8, W, d_rec.p + nd * (W + 1),
                                    D - nd, d_cnt, s));
    }
  }
  // ---- 2. hash -> code -------------------------------------------------------------------------
  const MaterializedColumn& m = t->materialized[cl.name];
  uint64_t cap = 1024;
  while (cap < 2 * D) cap <<= 1;
  DevBuf<uint64_t> d_tab, d_special;
  DevBuf<uint32_t> d_status;
  DICT_HIP(d_tab.alloc(cap * 2 * 8));
  DICT_HIP(d_special.alloc(16));
  DICT_HIP(d_status.alloc(16));
  DICT_HIP(hipMemsetAsync(d_tab, 0xff, cap * 2 * 8, s));
  DICT_HIP(hipMemsetAsync(d_special, 0xff, 16, s));
  DICT_HIP(hipMemsetAsync(d_status, 0, 16, s));
  DICT_HIP(hipMalloc(reinterpret_cast<void**>(&d->d_entries), D * 3 * 8));
  // codes: 4 bytes per row, laid out as 512 KiB "pages" of 131,072 values (the PLAIN32
  // accessor of the fused kernel), zero slack for the tile overhang
  const uint64_t per_page = 131072;
  const uint64_t npages = (n + per_page - 1) / per_page;
  const uint64_t bytes = npages * per_page * 4 + (1 << 20);
  DICT_HIP(hipMalloc(reinterpret_cast<void**>(&d->d_codes), bytes));
  DICT_HIP(hipMemsetAsync(d->d_codes, 0, bytes, s));
  std::vector<uint64_t> offs;
  for (uint64_t p = 0; p < npages; ++p) offs.push_back(p * per_page * 4);
  offs.push_back(offs.back());
  DICT_HIP(hipMalloc(reinterpret_cast<void**>(&d->d_code_pages), offs.size() * 8));
  DICT_HIP(hipMemcpyAsync(d->d_code_pages, offs.data(), offs.size() * 8, hipMemcpyHostToDevice, s));
  DictArgs da{};
  da.records = d_rec;
  da.ncodes = D;
  da.entries = d->d_entries;
  da.tab = d_tab;
  da.cap = cap;
  da.special = d_special;
  da.image = t->d_image;
  da.pages = t->d_pages[li][0];
  da.hashes = m.d_values;
  da.strpos = m.d_strpos;
  da.tags = cl.dlevel_max > 0 ? m.d_tags : nullptr;
  da.nrows = n;
  da.codes = d->d_codes;
  da.status = d_status;
  DICT_HIP(launch_dict_insert(da, s));
  // ---- 3. codes + the proof of exactness --------------------------------------------------------
  DICT_HIP(launch_dict_assign(da, s));
  uint32_t status[4] = {0, 0, 0, 0};
  DICT_HIP(hipMemcpyAsync(status, d_status, 16, hipMemcpyDeviceToHost, s));
  DICT_HIP(hipStreamSynchronize(s));
  if (status[0] != 0) {
    d->why = status[0] & 2u ? "two different strings share a 64-bit hash"
                            : "a row's hash is missing from the dictionary";
    return Status();
  }
  d->n_codes = D;
  d->usable = true;
  return Status();
}

Status table_string_dict(evql_table* t, int li, StringDict** out) {
  const std::string& name = t->layout.columns[li].name;
  StringDict& d = t->dicts[name];
  if (!d.tried) {
    d.tried = true;
    Status st = build_dict(t, li, &d);
    if (!st.ok()) {
      d.usable = false;
      d.why = st.msg;
      // (a device error while building is an error of the query that asked)
      if (st.code == EVQL_EDEVICE) return st;
    }
    if (!d.usable) {
      if (d.d_codes) hipFree(d.d_codes);
      if (d.d_code_pages) hipFree(d.d_code_pages);
      if (d.d_entries) hipFree(d.d_entries);
      d.d_codes = nullptr;
      d.d_code_pages = nullptr;
      d.d_entries = nullptr;
    }
  }
  *out = &d;
  return Status();
}

}  // namespace evql
