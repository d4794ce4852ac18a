// merge.cc -- GroupByMergeExpression for PartialGroupBy wire rows.
//
// Reference: sql/statements/select/groupby.cc:528-637 (result_handler: per row a
// 20-byte SHA1 group key, then per select expression either the aggregate's
// saved state -> loadInstanceState + mergeInstance, or an SValue::decode that
// overwrites the group's value) and :640-672 (nextBatch: method_call per select
// expression / copyBoxed).  Frame: varuint flags, varuint count, rows
// (native_transport/frames/query_partialaggr_result.cc:53-57).
//
// This is the coordinator's side of the exchange for partial aggregates that
// arrive as bytes (from evql_query_next_batch in EVQL_MODE_PARTIAL on another
// node, or from an unmodified reference node): its work is O(groups x frames) of
// sequential LEB128 decoding, not a scan, so it stays on the host.  Partial
// tables that live on other GPUs are merged on the device instead
// (evql_query_import_groups).
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <unordered_map>
#include "plan_ir.h"
#include "runtime.h"

using namespace evql;

namespace {

struct KeyHash {
  // util/SHA1.h:104-111: first 8 bytes of the digest
  size_t operator()(const std::string& k) const {
    uint64_t h = 0;
    memcpy(&h, k.data(), k.size() < 8 ? k.size() : 8);
    return size_t(h);
  }
};

struct Cell {
  uint64_t w0 = 0;   // aggregate state / numeric SValue payload
  uint64_t cnt = 0;  // min / max / mean: non-null inputs seen
  std::string str;   // non-aggregate string payload
  std::unique_ptr<std::set<uint64_t>> dset;  // count_distinct (aggregate.cc:77-80)
  uint8_t tag = 0;
  uint32_t type = EVQL_T_NIL;  // non-aggregates: decoded SType
};

struct Reader {
  const uint8_t* p;
  size_t len, pos = 0;
  bool ok = true;
  uint64_t varuint() {
    // util/io/inputstream.cc readVarUInt: 7 bits per byte, LSB group first
    uint64_t v = 0;
    for (int i = 0; i < 10; ++i) {
      if (pos >= len) {
        ok = false;
        return 0;
      }
      const uint8_t b = p[pos++];
      v |= uint64_t(b & 0x7f) << (7 * i);
      if (!(b & 0x80)) return v;
    }
    return v;
  }
  const uint8_t* bytes(size_t n) {
    if (n > len - pos) {
      ok = false;
      return nullptr;
    }
    const uint8_t* r = p + pos;
    pos += n;
    return r;
  }
};

}  // namespace

struct evql_merge {
  std::vector<LoweredProgram> select;
  std::unordered_map<std::string, std::vector<Cell>, KeyHash> groups;
  std::unordered_map<std::string, std::vector<Cell>, KeyHash>::const_iterator it;
  bool iterating = false;
  uint64_t frames = 0, rows = 0;
  std::vector<std::vector<uint8_t>> out_cols;
};

namespace {

// instance_loadstate, inverse of save_state() in runtime.cc
bool load_state(uint32_t fn, Reader* r, Cell* c) {
  switch (fn) {
    case EVQL_AGG_COUNT:
    case EVQL_AGG_SUM_UINT64:
    case EVQL_AGG_SUM_INT64:
      c->w0 = r->varuint();
      c->cnt = 0;
      return r->ok;
    case EVQL_AGG_SUM_FLOAT64: {
      const uint8_t* b = r->bytes(8);
      if (b) memcpy(&c->w0, b, 8);
      return r->ok;
    }
    case EVQL_AGG_COUNT_DISTINCT_UINT64: {  // aggregate.cc:119-125
      const uint64_t n = r->varuint();
      c->dset.reset(new std::set<uint64_t>());
      for (uint64_t i = 0; i < n && r->ok; ++i) c->dset->insert(r->varuint());
      return r->ok;
    }
    default: {
      c->cnt = r->varuint();
      const uint8_t* b = r->bytes(8);
      if (b) memcpy(&c->w0, b, 8);
      return r->ok;
    }
  }
}

template <typename T>
void fold_minmax(bool is_min, Cell* self, const Cell& o) {
  if (o.cnt == 0) return;
  T a, b;
  memcpy(&b, &o.w0, 8);
  if (self->cnt == 0) {
    a = b;
  } else {
    memcpy(&a, &self->w0, 8);
    if (is_min ? (b < a) : (b > a)) a = b;
  }
  memcpy(&self->w0, &a, 8);
  self->cnt += o.cnt;
}

// instance_merge (aggregate.cc:47-49,163-165,199-201 for the reference's own)
void merge_state(uint32_t fn, Cell* self, const Cell& o) {
  switch (fn) {
    case EVQL_AGG_COUNT:
    case EVQL_AGG_SUM_UINT64:
    case EVQL_AGG_SUM_INT64:
      self->w0 += o.w0;
      return;
    case EVQL_AGG_COUNT_DISTINCT_UINT64:  // aggregate.cc:103-109
      if (o.dset) {
        if (!self->dset) self->dset.reset(new std::set<uint64_t>());
        self->dset->insert(o.dset->begin(), o.dset->end());
      }
      return;
    case EVQL_AGG_SUM_FLOAT64:
    case EVQL_AGG_MEAN_UINT64:
    case EVQL_AGG_MEAN_INT64:
    case EVQL_AGG_MEAN_FLOAT64: {
      double a, b;
      memcpy(&a, &self->w0, 8);
      memcpy(&b, &o.w0, 8);
      a += b;
      memcpy(&self->w0, &a, 8);
      self->cnt += o.cnt;
      return;
    }
    case EVQL_AGG_MIN_UINT64: fold_minmax<uint64_t>(true, self, o); return;
    case EVQL_AGG_MAX_UINT64: fold_minmax<uint64_t>(false, self, o); return;
    case EVQL_AGG_MIN_INT64: fold_minmax<int64_t>(true, self, o); return;
    case EVQL_AGG_MAX_INT64: fold_minmax<int64_t>(false, self, o); return;
    case EVQL_AGG_MIN_FLOAT64: fold_minmax<double>(true, self, o); return;
    case EVQL_AGG_MAX_FLOAT64: fold_minmax<double>(false, self, o); return;
  }
}

// instance `get`: what X_CALL_INSTANCE pushes in method_call
Value state_value(uint32_t fn, const Cell& c) {
  Value v;
  v.tag = 0;
  v.bits = c.w0;
  switch (fn) {
    case EVQL_AGG_COUNT:
    case EVQL_AGG_SUM_UINT64: v.type = EVQL_T_UINT64; return v;
    case EVQL_AGG_COUNT_DISTINCT_UINT64:
      v.type = EVQL_T_UINT64;
      v.bits = c.dset ? c.dset->size() : 0;
      return v;
    case EVQL_AGG_SUM_INT64: v.type = EVQL_T_INT64; return v;
    case EVQL_AGG_SUM_FLOAT64: v.type = EVQL_T_FLOAT64; return v;
    case EVQL_AGG_MIN_UINT64:
    case EVQL_AGG_MAX_UINT64: v.type = EVQL_T_UINT64; break;
    case EVQL_AGG_MIN_INT64:
    case EVQL_AGG_MAX_INT64: v.type = EVQL_T_INT64; break;
    case EVQL_AGG_MIN_FLOAT64:
    case EVQL_AGG_MAX_FLOAT64: v.type = EVQL_T_FLOAT64; break;
    default: {
      v.type = EVQL_T_FLOAT64;
      if (c.cnt) {
        double s;
        memcpy(&s, &c.w0, 8);
        s /= double(c.cnt);
        memcpy(&v.bits, &s, 8);
      }
    }
  }
  if (c.cnt == 0) {
    v.bits = 0;
    v.tag = EVQL_STAG_NULL;
  }
  return v;
}

// SValue::decode (svalue.cc:311-315): u8 type, lenenc(value bytes || tag)
bool decode_svalue(Reader* r, Cell* c) {
  const uint8_t* t = r->bytes(1);
  if (!t) return false;
  c->type = *t;
  const uint64_t n = r->varuint();
  const uint8_t* d = r->ok ? r->bytes(n) : nullptr;
  if (!d) return false;
  c->w0 = 0;
  c->str.clear();
  c->tag = 0;
  switch (c->type) {
    case EVQL_T_NIL:
      c->tag = n ? d[n - 1] : uint8_t(EVQL_STAG_NULL);
      return true;
    case EVQL_T_BOOL:
      if (n != 2) return false;
      c->w0 = d[0];
      c->tag = d[1];
      return true;
    case EVQL_T_STRING: {
      if (n < 5) return false;
      uint32_t sl;
      memcpy(&sl, d, 4);
      if (uint64_t(sl) + 5 != n) return false;
      c->str.assign(reinterpret_cast<const char*>(d + 4), sl);
      c->tag = d[n - 1];
      return true;
    }
    case EVQL_T_UINT64:
    case EVQL_T_INT64:
    case EVQL_T_FLOAT64:
    case EVQL_T_TIMESTAMP64:
      if (n != 9) return false;
      memcpy(&c->w0, d, 8);
      c->tag = d[8];
      return true;
    default:
      return false;
  }
}

Status merge_rows(evql_merge* m, Reader* r, uint64_t count) {
  const size_t nsel = m->select.size();
  Cell remote;
  for (uint64_t j = 0; j < count; ++j) {
    const uint8_t* k = r->bytes(20);
    if (!k) return Status::error(EVQL_EIO, "invalid partialaggr result encoding");
    auto& group = m->groups[std::string(reinterpret_cast<const char*>(k), 20)];
    if (group.empty()) {
      group.resize(nsel);
      for (size_t i = 0; i < nsel; ++i) {
        if (!m->select[i].is_aggregate) {
          group[i].type = m->select[i].return_type;
          group[i].tag = EVQL_STAG_NULL;
        }
      }
    }
    for (size_t i = 0; i < nsel; ++i) {
      const LoweredProgram& lp = m->select[i];
      if (lp.is_aggregate) {
        remote = Cell();
        if (!load_state(lp.aggregate_fn, r, &remote)) {
          return Status::error(EVQL_EIO, "invalid partialaggr result encoding");
        }
        merge_state(lp.aggregate_fn, &group[i], remote);
      } else if (!decode_svalue(r, &group[i])) {
        return Status::error(EVQL_EIO, "invalid partialaggr result encoding");
      }
    }
    ++m->rows;
  }
  m->iterating = false;
  return Status();
}

}  // namespace

extern "C" {

int evql_merge_create(const evql_plan_desc_t* plan, evql_merge_t** out) {
  if (!plan || !out) return fail(EVQL_EARG, "null argument");
  std::unique_ptr<evql_merge> m(new evql_merge());
  for (uint32_t i = 0; i < plan->n_select; ++i) {
    LoweredProgram lp;
    bool unsup = false;
    std::string e = lower_program(plan->select_exprs[i], &lp, &unsup);
    if (!e.empty()) return fail(unsup ? EVQL_ENOTSUP : EVQL_EARG, e);
    m->select.push_back(lp);
  }
  *out = m.release();
  return EVQL_OK;
}

void evql_merge_destroy(evql_merge_t* m) { delete m; }

int evql_merge_add_frame(evql_merge_t* m, const void* payload, size_t len) {
  if (!m || (!payload && len)) return fail(EVQL_EARG, "null argument");
  Reader r{static_cast<const uint8_t*>(payload), len};
  r.varuint();  // flags
  const uint64_t count = r.varuint();
  if (!r.ok) return fail(EVQL_EIO, "invalid partialaggr result encoding");
  Status st = merge_rows(m, &r, count);
  if (!st.ok()) return fail(st.code, st.msg);
  ++m->frames;
  return EVQL_OK;
}

int evql_merge_add_rows(evql_merge_t* m, const void* keys, size_t keys_len, const void* data,
                        size_t data_len, size_t nrows) {
  if (!m) return fail(EVQL_EARG, "null argument");
  // two STRING SVectors (u32 len, bytes, tag per element), as PartialGroupBy's
  // nextBatch hands them to its parent (groupby.cc:438-472)
  Reader kr{static_cast<const uint8_t*>(keys), keys_len};
  Reader dr{static_cast<const uint8_t*>(data), data_len};
  std::vector<uint8_t> row;
  for (size_t i = 0; i < nrows; ++i) {
    const uint8_t* kl = kr.bytes(4);
    const uint8_t* dl = dr.bytes(4);
    if (!kl || !dl) return fail(EVQL_EIO, "invalid partial aggregate rows");
    uint32_t kn, dn;
    memcpy(&kn, kl, 4);
    memcpy(&dn, dl, 4);
    const uint8_t* kb = kr.bytes(size_t(kn) + 1);
    const uint8_t* db = dr.bytes(size_t(dn) + 1);
    if (!kb || !db || kn != 20) return fail(EVQL_EIO, "invalid partial aggregate rows");
    row.assign(kb, kb + 20);
    row.insert(row.end(), db, db + dn);
    Reader r{row.data(), row.size()};
    Status st = merge_rows(m, &r, 1);
    if (!st.ok()) return fail(st.code, st.msg);
    if (r.pos != row.size()) return fail(EVQL_EIO, "invalid partial aggregate rows");
  }
  return EVQL_OK;
}

uint64_t evql_merge_num_groups(const evql_merge_t* m) { return m->groups.size(); }
size_t evql_merge_column_count(const evql_merge_t* m) { return m->select.size(); }
uint32_t evql_merge_column_type(const evql_merge_t* m, size_t idx) {
  return idx < m->select.size() ? m->select[idx].return_type : uint32_t(EVQL_T_NIL);
}

int evql_merge_next_batch(evql_merge_t* m, size_t max_rows, evql_column_buf_t* cols,
                          size_t* nrows) {
  if (!m || !cols || !nrows) return fail(EVQL_EARG, "null argument");
  if (!m->iterating) {
    m->it = m->groups.begin();
    m->iterating = true;
  }
  const size_t nsel = m->select.size();
  m->out_cols.assign(nsel, std::vector<uint8_t>());
  size_t emitted = 0;
  const std::vector<Value> no_inputs;
  while (m->it != m->groups.end() && emitted < max_rows) {
    const std::vector<Cell>& g = m->it->second;
    for (size_t i = 0; i < nsel; ++i) {
      const LoweredProgram& lp = m->select[i];
      Value out;
      if (lp.is_aggregate) {
        const Value av = state_value(lp.aggregate_fn, g[i]);
        std::string e = eval_expr(lp.call, no_inputs, &av, &out);
        if (!e.empty()) return fail(EVQL_ERUNTIME, e);
      } else {
        // copyBoxed of the decoded SValue (groupby.cc:660-662)
        out.type = lp.return_type;
        out.bits = g[i].w0;
        out.str = g[i].str;
        out.tag = g[i].tag;
      }
      append_svector(lp.return_type, out, &m->out_cols[i]);
    }
    ++m->it;
    ++emitted;
  }
  for (size_t i = 0; i < nsel; ++i) {
    cols[i].data = m->out_cols[i].data();
    cols[i].size = m->out_cols[i].size();
  }
  *nrows = emitted;
  return EVQL_OK;
}

}  // extern "C"
