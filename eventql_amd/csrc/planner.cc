// planner.cc -- from evql_plan_desc_t (bytecode of the reference's operators)
// to a KernelPlan: which pages each column is read from, the inlined
// predicate / key / aggregate-argument expressions and the group-table layout.
//
// Mirrors the plan shape the reference builds for
//   GroupByExpression(select_exprs, group_exprs, FastCSTableScan(stmt))
// (sql/scheduler.cc:134-182): scan-level programs index scan columns,
// group-level programs index the scan's select list.
#include <cstdlib>
#include <cstring>
#include "runtime.h"

namespace evql {

namespace {

// replace INPUT(j) by a copy of sub[j]
ExprPtr inline_inputs(const ExprPtr& e, const std::vector<ExprPtr>& sub, std::string* err) {
  if (!e) return e;
  if (e->kind == Expr::INPUT) {
    if (e->input >= sub.size()) {
      *err = "invalid input index";
      return e;
    }
    return sub[e->input];
  }
  auto c = std::make_shared<Expr>(*e);
  for (auto& a : c->args) a = inline_inputs(a, sub, err);
  return c;
}

bool has_agg_get(const ExprPtr& e) {
  if (!e) return false;
  if (e->kind == Expr::AGG_GET) return true;
  for (const auto& a : e->args) {
    if (has_agg_get(a)) return true;
  }
  return false;
}

bool has_input(const ExprPtr& e) {
  if (!e) return false;
  if (e->kind == Expr::INPUT) return true;
  for (const auto& a : e->args) {
    if (has_input(a)) return true;
  }
  return false;
}

bool is_bare_input(const ExprPtr& e) { return e && e->kind == Expr::INPUT; }

bool is_string_compare(const ExprPtr& e) {
  if (e->kind != Expr::CALL || e->type_slot != EVQL_TS_STRING) return false;
  return (e->family >= EVQL_FAM_CMP && e->family <= EVQL_FAM_GTE) ||
         e->family == EVQL_FAM_STARTSWITH || e->family == EVQL_FAM_ENDSWITH;
}

// Strings on the device: a bare column reference flows through untouched (as
// its hash, for keys / first-row values), and string columns / literals may be
// the operands of eq / neq / lt / lte / gt / gte / cmp / startswith / endswith (bytewise
// compare in the kernel).  Anything else that produces or consumes a string (concat, IF over
// strings, to_string ...) is not lowered.
bool strings_lowerable(const ExprPtr& e, bool root = true) {
  if (!e) return true;
  if (e->type == EVQL_T_STRING) return root && e->kind == Expr::INPUT;
  if (is_string_compare(e)) {
    for (const auto& a : e->args) {
      if (a->type != EVQL_T_STRING) return false;
      if (a->kind != Expr::INPUT && a->kind != Expr::LITERAL) return false;
    }
    return true;
  }
  for (const auto& a : e->args) {
    if (!strings_lowerable(a, false)) return false;
  }
  return true;
}

// A predicate that holds for every row whatever the data: what the reference's planner
// leaves in place and evaluates per row, e.g. `x >= 0` over an unsigned column (NULL
// compares as 0, vm.cc:231-272), `(s = '') OR (s != '')`, `(f = true) OR (f = false)`.
// Operands are columns and literals only (nothing that could raise).  Dropping such a
// WHERE changes no result; for nested scans it also removes the only way a row could be
// rejected, and with it the reference's value resets behind rejected rows
// (CSTableScan.cc:501-515).
bool simple_operand(const ExprPtr& e) { return e->kind == Expr::INPUT || e->kind == Expr::LITERAL; }

bool expr_always_true(const ExprPtr& e) {
  if (!e) return false;
  if (e->kind == Expr::LITERAL) return e->type == EVQL_T_BOOL && e->lit_bits != 0;
  if (e->kind != Expr::CALL) return false;
  const auto& a = e->args;
  switch (e->family) {
    case EVQL_FAM_LOGICAL_AND:
      return expr_always_true(a[0]) && expr_always_true(a[1]);
    case EVQL_FAM_LOGICAL_OR: {
      if (expr_always_true(a[0]) || expr_always_true(a[1])) return true;
      if (a[0]->kind != Expr::CALL || a[1]->kind != Expr::CALL) return false;
      const ExprPtr &l = a[0], &r = a[1];
      if (l->args.size() != 2 || r->args.size() != 2) return false;
      if (!simple_operand(l->args[0]) || !simple_operand(l->args[1])) return false;
      const bool same = expr_equal(l->args[0], r->args[0]) && expr_equal(l->args[1], r->args[1]);
      // x = y OR x != y  (either order; true for NaN as well: != holds)
      if (same && ((l->family == EVQL_FAM_EQ && r->family == EVQL_FAM_NEQ) ||
                   (l->family == EVQL_FAM_NEQ && r->family == EVQL_FAM_EQ))) {
        return true;
      }
      // b = true OR b = false over a BOOL column (values are normalised to 0 / 1)
      if (l->family == EVQL_FAM_EQ && r->family == EVQL_FAM_EQ && l->type_slot == EVQL_TS_BOOL &&
          r->type_slot == EVQL_TS_BOOL && expr_equal(l->args[0], r->args[0]) &&
          l->args[0]->kind == Expr::INPUT && l->args[1]->kind == Expr::LITERAL &&
          r->args[1]->kind == Expr::LITERAL && (l->args[1]->lit_bits != 0) != (r->args[1]->lit_bits != 0)) {
        return true;
      }
      return false;
    }
    case EVQL_FAM_GTE:  // x >= 0, unsigned
      return (e->type_slot == EVQL_TS_UINT64 || e->type_slot == EVQL_TS_TIMESTAMP64) &&
             simple_operand(a[0]) && a[1]->kind == Expr::LITERAL && a[1]->lit_bits == 0;
    case EVQL_FAM_LTE:  // 0 <= x
      return (e->type_slot == EVQL_TS_UINT64 || e->type_slot == EVQL_TS_TIMESTAMP64) &&
             simple_operand(a[1]) && a[0]->kind == Expr::LITERAL && a[0]->lit_bits == 0;
    default:
      return false;
  }
}

// marks the string columns whose bytes the kernel has to reach
void mark_string_bytes(const ExprPtr& e, std::vector<ColAccess>* cols) {
  if (!e) return;
  if (is_string_compare(e)) {
    for (const auto& a : e->args) {
      if (a->kind == Expr::INPUT && a->input < cols->size()) (*cols)[a->input].string_bytes = true;
    }
  }
  for (const auto& a : e->args) mark_string_bytes(a, cols);
}

// can the value carry STAG_NULL?  (pure functions always produce tag 0)
bool expr_may_be_null(const ExprPtr& e, const std::vector<ColAccess>& cols) {
  if (!e) return false;
  switch (e->kind) {
    case Expr::INPUT:
      return e->input < cols.size() && cols[e->input].has_tags;
    case Expr::LITERAL:
      return (e->lit_tag & 1) != 0;
    case Expr::IF:
      return expr_may_be_null(e->args[1], cols) || expr_may_be_null(e->args[2], cols);
    default:
      return false;
  }
}

int op_for_minmax(uint32_t fn) {
  switch (fn) {
    case EVQL_AGG_MIN_UINT64: return 2;
    case EVQL_AGG_MAX_UINT64: return 3;
    case EVQL_AGG_MIN_INT64: return 4;
    case EVQL_AGG_MAX_INT64: return 5;
    case EVQL_AGG_MIN_FLOAT64: return 6;
    case EVQL_AGG_MAX_FLOAT64: return 7;
  }
  return 0;
}

}  // namespace

// ---- launch shape ---------------------------------------------------------------------
// Chooses block size, unroll, LDS table size and -- for `hint` groups far beyond
// what an LDS table holds -- the partitioned path.  Called when the plan is built
// (hint = evql_plan_desc_t::groups_hint) and again by the runtime when a plan
// without a hint turns out to be high-cardinality (query_launch's sample pass).
uint64_t lds_table_max_slots(const KernelPlan& kp) {
  const uint64_t W = uint64_t(kp.words_per_slot());
  uint64_t smax = 256;
  while ((smax * 2 + 2) * 8 * W <= 150 * 1024) smax <<= 1;
  return smax;
}

bool partitioned_path_possible(const KernelPlan& kp) {
  // (a nullable exact key needs the NULL slot; count_distinct inserts into its pair
  // set from the row function, which the partition passes would run twice)
  return kp.key_mode != KEY_NONE &&
         !(kp.key_mode == KEY_EXACT && expr_may_be_null(kp.group[0], kp.cols)) &&
         kp.n_distinct == 0;
}

// the partitioned path takes over as soon as the expected groups exceed the LDS slots
const uint64_t kPartitionAboveSlots = 1;

void choose_launch_shape(KernelPlan* kpp, uint64_t hint) {
  KernelPlan& kp = *kpp;
  kp.block = 256;
  kp.partitioned = false;
  kp.lane_cache = 1;
  // loads in flight per lane = columns x unroll; ~16 saturate HBM (measured: 2
  // columns 2.54 ms at unroll 4, 2.43 ms at unroll 8; 4 columns spill at 8)
  kp.unroll = kp.cols.size() <= 2 ? 8 : 4;
  {
    // A 1024-thread workgroup leaves each wave 128 VGPRs.  Rough budget: ~50 for the
    // loop itself, 4 (+1 with tags) per column and unroll step for the tile, 4 per update
    // word (row outputs + the lane-private run accumulators).  Plans beyond it start with
    // fewer unroll steps; the runtime halves again if the compiled kernel still reports
    // scratch memory (compile_plan_kernels) -- measured on a 12-aggregate plan over
    // config 3's table: 2.66 ms spilling at unroll 4, 1.88 ms at unroll 1.
    int upd = 0;
    for (const auto& a : kp.aggs) upd += a.nwords;
    int per_step = 0;
    for (const auto& c : kp.cols) per_step += c.has_tags ? 5 : 4;
    // (the estimate runs ~15 high: config 3 -- 4 columns, 3 update words -- compiles to 109)
    while (kp.unroll > 1 && 50 + per_step * kp.unroll + 4 * upd > 144) kp.unroll /= 2;
  }
  // (narrow re-encoded columns at unroll 8 spill: config3l 0.39 ms at 4, 0.72 ms at 8)
  kp.lds_slots = 0;
  if (kp.key_mode == KEY_NONE) return;
  // One 1024-thread workgroup per CU owning (almost) the whole 160 KiB LDS: the
  // probe loop of a wave runs as long as its slowest lane, so the table is kept
  // sparse (load factor <= 1/4 where the LDS allows) -- measured on MI355X:
  // 1000 groups in 4096 slots ran 2.1x faster than in 2048 slots.
  const uint64_t smax = lds_table_max_slots(kp);
  // Dense keys are placed by identity, so even `hint` close to the slot count
  // works (4000 dense groups in 4096 slots: 0.63 ms vs 24 ms HBM-only for 2e8
  // rows); a workgroup whose table does thrash switches itself to the HBM
  // table (`bypass`).  Only far beyond the LDS capacity is the table skipped.
  if (hint > kPartitionAboveSlots * smax && partitioned_path_possible(kp)) {
    // more groups than LDS slots: the rows of the groups that find no slot would go to
    // the HBM table with one random atomic per state word, which tops out at the
    // chip's scattered-atomic rate (~2e10/s measured) -- with 3 state words, 13 % of
    // the rows overflowing already cost as much as partitioning all of them (3 tuple
    // passes at ~5 TB/s), and a plan with 20 state words whose 1000 groups met a
    // 256-slot table ran 346 ms per 2e8 rows.  Instead the passing
    // rows are radix-partitioned into buckets small enough for the LDS table
    // (streaming traffic) and every bucket is aggregated in LDS.
    kp.partitioned = true;
    kp.lds_slots = int(smax);
    kp.block = 1024;
    kp.part_bits = 8;
    while (kp.part_bits < 14 && (hint >> kp.part_bits) > smax / 2) ++kp.part_bits;
    // (measured on config 4, 1e7 groups, 16-byte tuples: 4096 slots / 13 bits 3.08 ms;
    //  12 bits 3.24 ms; 2048 slots so that two aggregate workgroups share a CU: 13 bits
    //  3.24 ms, 14 bits 3.17 ms -- the sparse table beats the overlap)
  } else if (hint > 8 * smax) {
    kp.lds_slots = 0;  // (nullable exact key) aggregate straight into the HBM table
    kp.block = 256;
  } else {
    uint64_t s = smax;
    if (hint != 0) {
      s = 1024;
      while (s < hint * 4) s <<= 1;
      if (s > smax) s = smax;
    }
    kp.lds_slots = int(s);
    kp.block = 1024;
    // 2 .. 4 groups: half, a third, a quarter of a wave's rows meet on ONE LDS address and
    // the atomics serialise (config 2's query, 2e8 rows: 1.03 / 0.82 / 0.71 ms for 2 / 3 / 4
    // groups against 0.50 ms for 9 and 0.56 ms for 1000).  Four lane-private accumulators
    // then hold every group a lane meets (codegen_kernels.inc evql_update).
    kp.lane_cache = (hint >= 2 && hint <= 4) ? 4 : 1;
    // (measured, round 2: a 2048-slot table runs dense keys as fast as 4096 slots, but the
    //  freed LDS does not buy a second workgroup per CU: at 64 VGPRs per wave the kernel
    //  spills -- config 3 over 16-bit pages 0.38 -> 1.03 ms; with unroll 2 on top 0.50 ms,
    //  10-bit config 2 2.18 -> 2.52 ms, config 5 0.54 -> 0.59 ms)
  }
}

Status build_kernel_plan(const TableLayout& layout, const evql_plan_desc_t* plan,
                         evql_query* q, bool* unsupported) {
  *unsupported = false;
  KernelPlan& kp = q->kp;
  auto unsup = [&](const std::string& m) {
    *unsupported = true;
    return Status::error(EVQL_ENOTSUP, m);
  };

  const bool within = plan->scan_mode == EVQL_SCAN_NESTED_WITHIN_RECORD;
  const bool nested = plan->scan_mode == EVQL_SCAN_NESTED || within;
  if (plan->scan_mode != EVQL_SCAN_FLAT && !nested) return Status::error(EVQL_EARG, "bad scan mode");
  q->nested = nested;
  q->within_record = within;
  if (within && plan->where) return unsup("WHERE in a WITHIN RECORD scan is not lowered");
  if (within && plan->n_scan_select == 0) {
    return Status::error(EVQL_EARG, "WITHIN RECORD scan without a scan select list");
  }
  if (nested && plan->row_filter_bits) return unsup("row filter on a nested scan");
  if (nested && (plan->row_begin || plan->row_end)) return unsup("row range on a nested scan");
  if (plan->n_scan_columns > EVQL_MAX_COLS_HOST) return unsup("too many scan columns");
  if (plan->n_select == 0) return unsup("bare scans are not lowered (no GROUP BY / aggregate)");

  // ---- scan columns --------------------------------------------------------------
  for (uint32_t i = 0; i < plan->n_scan_columns; ++i) {
    ColAccess c;
    c.name = plan->scan_columns[i];
    c.stype = plan->scan_column_types[i];
    const ColumnLayout* cl = nullptr;
    for (size_t k = 0; k < layout.columns.size(); ++k) {
      if (layout.columns[k].name == c.name) {
        cl = &layout.columns[k];
        c.layout_index = int(k);
      }
    }
    if (!cl) {
      // the reference would dereference a null reader here (CSTableScan.cc:747-751)
      return Status::error(EVQL_EARG, "column not found: " + c.name);
    }
    // FastCSTableScan reads batch_size *values* of a repeated column and returns
    // wrong rows (SURVEY.md header); the reference routes such scans through
    // CSTableScan, i.e. EVQL_SCAN_NESTED
    if (cl->rlevel_max > 0 && !nested) return unsup("repeated column in a flat scan: " + c.name);
    switch (c.stype) {
      case EVQL_T_NIL:
        return Status::error(EVQL_EARG, "illegal column type: NIL");
      case EVQL_T_INT64:  // CSTableScan.cc:783-784
        return Status::error(EVQL_EARG, "illegal column type: INT64");
      default:
        break;
    }
    const bool is_uintish = cl->logical_type == ColumnType::UNSIGNED_INT ||
                            cl->logical_type == ColumnType::DATETIME ||
                            cl->logical_type == ColumnType::BOOLEAN;
    if (c.stype == EVQL_T_STRING) {
      if (cl->logical_type != ColumnType::STRING) return unsup("string view of a non-string column");
      c.mode = ColAccess::SOA;
      c.string_hash = true;
      c.has_tags = true;
    } else if (cl->logical_type == ColumnType::STRING) {
      return unsup("numeric view of a string column");
    } else if (c.stype == EVQL_T_FLOAT64) {
      if (cl->logical_type == ColumnType::FLOAT) {
        c.mode = ColAccess::PLAIN64;
      } else if (is_uintish) {
        c.from_uint_to_float = true;  // UnsignedIntColumnReader::readFloat casts
      } else {
        return unsup("unsupported column type");
      }
    } else {
      if (cl->logical_type == ColumnType::FLOAT) return unsup("integer view of a float column");
      if (!is_uintish) return unsup("unsupported column type");
      // readBoolean = value > 0 (column_reader_uint.cc:78-90)
      c.bool_normalize = c.stype == EVQL_T_BOOL;
    }
    if (!c.string_hash) {
      switch (cl->storage_type) {
        case ColumnEncoding::UINT64_PLAIN:
        case ColumnEncoding::FLOAT_IEEE754:
          c.mode = ColAccess::PLAIN64;
          break;
        case ColumnEncoding::UINT32_PLAIN:
          c.mode = ColAccess::PLAIN32;
          break;
        case ColumnEncoding::UINT32_BITPACKED:
        case ColumnEncoding::BOOLEAN_BITPACKED:
          c.mode = ColAccess::BITPACKED;
          c.bits = 0;  // filled in by the runtime from the page header
          break;
        case ColumnEncoding::UINT64_LEB128:
          c.mode = ColAccess::SOA;
          break;
        default:
          return unsup("unsupported column encoding");
      }
      if (cl->dlevel_max > 0) {
        c.mode = ColAccess::SOA;  // nullable: decoded to values + tags first
        c.has_tags = true;
      }
    }
    if (nested) {
      // CSTableScan path: every column is flattened to one value per output row;
      // undefined slots read as 0 with tag 0 (CSTableScan.cc:224-246)
      // (strings: hash + byte position per flattened row, materialize_nested)
      if (c.string_hash && within) return unsup("string columns in a record scan are not lowered");
      c.mode = ColAccess::SOA;
      c.has_tags = false;
    }
    kp.cols.push_back(c);
  }
  if (nested && !kp.cols.empty()) {
    // Every referenced column must lie on the leaf's ancestor chain: k_flatten_parent
    // indexes a shallower column's slots with counts taken from the LEAF's repetition
    // levels, which is meaningless for a column of a sibling repeated group (the
    // reference zips such groups).  The v0.2.0 header holds no tree,
    // so the chain is checked on the dotted names: the number of repeated nodes on
    // the common ancestor path P of a column A and the leaf is at most
    // min{rlevel_max(C) : C below P}; A's repeated ancestors are all on P only if
    // that bound reaches rlevel_max(A).  (materialize_nested adds an exact check on
    // the slot counts.)
    size_t leaf = 0;
    for (size_t i = 0; i < kp.cols.size(); ++i) {
      if (layout.columns[kp.cols[i].layout_index].rlevel_max >
          layout.columns[kp.cols[leaf].layout_index].rlevel_max) {
        leaf = i;
      }
    }
    const ColumnLayout& lc = layout.columns[kp.cols[leaf].layout_index];
    auto parent_path = [](const std::string& a, const std::string& b) {
      // longest common prefix of a and b that ends at a '.' of both ("" = root);
      // identical names share the whole name
      if (a == b) return a;
      size_t n = 0, last = 0;
      while (n < a.size() && n < b.size() && a[n] == b[n]) {
        if (a[n] == '.') last = n;
        ++n;
      }
      return a.substr(0, last);
    };
    for (const auto& c : kp.cols) {
      const ColumnLayout& cl = layout.columns[c.layout_index];
      if (cl.rlevel_max == 0 || &cl == &lc) continue;
      const std::string p = parent_path(cl.name, lc.name);
      uint32_t bound = ~0u;
      for (const auto& other : layout.columns) {
        const bool below = p.empty() || other.name == p ||
                           (other.name.size() > p.size() && other.name[p.size()] == '.' &&
                            other.name.compare(0, p.size(), p) == 0);
        if (below && other.rlevel_max < bound) bound = other.rlevel_max;
      }
      if (bound < cl.rlevel_max) {
        // sibling repeated groups: zipped level by level (runtime.cc materialize_nested_zip);
        // the record scan's per-record reduction assumes one chain
        if (within) {
          return unsup("nested columns from different repeated groups in a record scan: " + cl.name +
                       ", " + lc.name);
        }
        q->nested_siblings = true;
      }
    }
  }
  bool where_always_true = false;
  if (nested && plan->where && !kp.cols.empty()) {
    LoweredProgram w;
    bool wu = false;
    if (lower_program(*plan->where, &w, &wu).empty()) where_always_true = expr_always_true(w.call);
  }
  if (nested && plan->where && !where_always_true) {
    // after a row rejected by WHERE the reference resets parent values without
    // re-reading them (CSTableScan.cc:501-512); only leaf-level-only scans are
    // free of that history dependence
    uint32_t lo = ~0u, hi = 0;
    for (const auto& c : kp.cols) {
      uint32_t rm = layout.columns[c.layout_index].rlevel_max;
      lo = rm < lo ? rm : lo;
      hi = rm > hi ? rm : hi;
    }
    if (!kp.cols.empty() && lo != hi) {
      // reproduced by the runtime (apply_where_resets): a parent value reads 0 behind
      // the first row of its slot when that row was rejected
      if (within) return unsup("WHERE over columns of different repetition depth in a record scan");
      if (q->nested_siblings) {
        return unsup("WHERE over columns of different repetition depth from sibling repeated groups");
      }
      kp.where_rows_kernel = true;
      q->nested_where_mixed = true;
    }
  }

  // ---- programs ----------------------------------------------------------------------
  bool u = false;
  std::string err;
  if (plan->where) {
    err = lower_program(*plan->where, &q->where, &u);
    if (!err.empty()) return u ? unsup(err) : Status::error(EVQL_EARG, err);
    if (q->where.is_aggregate || q->where.return_type != EVQL_T_BOOL) {
      return Status::error(EVQL_EARG, "WHERE must be a non-aggregate boolean expression");
    }
    if (!strings_lowerable(q->where.call)) return unsup("string expression is not lowerable");
    q->has_where = true;
    kp.where = q->where.call;
    if (where_always_true) {
      q->has_where = false;  // holds for every row: no row is rejected, nothing to evaluate
      kp.where = nullptr;
    }
    if (nested && kp.cols.empty()) {
      // CSTableScan::fetchNextWithoutColumns (CSTableScan.cc:551-564) skips the
      // record when the predicate is TRUE (`if (popBool(..)) continue;`): a scan
      // that reads no column emits its rows exactly when WHERE is false
      auto neg = std::make_shared<Expr>();
      neg->kind = Expr::CALL;
      neg->type = EVQL_T_BOOL;
      neg->family = EVQL_FAM_NEG;
      neg->type_slot = EVQL_TS_BOOL;
      neg->args.push_back(kp.where);
      kp.where = neg;
    }
    mark_string_bytes(kp.where, &kp.cols);
  }
  q->scan_select.resize(plan->n_scan_select);
  std::vector<ExprPtr> scan_out;
  if (within) {
    // the record scan reads the plan's columns; the operators above it read one
    // per-record value per scan select expression
    q->wr_cols.swap(kp.cols);
    // (without columns the reference takes fetchNextWithoutColumns and calls the
    // aggregates' `get` on a null instance, CSTableScan.cc:567-577)
    if (q->wr_cols.empty()) return Status::error(EVQL_EARG, "WITHIN RECORD scan without columns");
  }
  for (uint32_t i = 0; i < plan->n_scan_select && within; ++i) {
    LoweredProgram& lp = q->scan_select[i];
    err = lower_program(plan->scan_select[i], &lp, &u);
    if (!err.empty()) return u ? unsup(err) : Status::error(EVQL_EARG, err);
    if (!lp.is_aggregate || lp.call->kind != Expr::AGG_GET) {
      return unsup("WITHIN RECORD select expressions other than a bare aggregate");
    }
    evql_query::WithinAgg w;
    ExprPtr arg;
    switch (lp.aggregate_fn) {
      case EVQL_AGG_COUNT:
        w.is_count = true;
        if (lp.acc_args.size() == 1) {
          arg = lp.acc_args[0];
          if (arg->kind == Expr::CALL && arg->family == EVQL_FAM_TO_NIL) arg = arg->args[0];
        }
        break;
      case EVQL_AGG_SUM_UINT64:
      case EVQL_AGG_SUM_INT64:
        if (lp.acc_args.size() != 1) return Status::error(EVQL_EARG, "aggregate arity");
        arg = lp.acc_args[0];
        break;
      default:
        return unsup("WITHIN RECORD aggregates other than count / integer sum");
    }
    if (arg && arg->kind == Expr::INPUT) {
      if (arg->input >= q->wr_cols.size()) return Status::error(EVQL_EARG, "bad input index");
      const ColAccess& c = q->wr_cols[arg->input];
      if (!w.is_count && c.stype != EVQL_T_UINT64) {
        return unsup("WITHIN RECORD sum over a column that is not UINT64");
      }
      w.col = int(arg->input);
      w.level = layout.columns[c.layout_index].rlevel_max;
    } else if (arg && arg->kind == Expr::LITERAL) {
      if (!w.is_count && (arg->lit_tag != 0 ||
                          (arg->type != EVQL_T_UINT64 && arg->type != EVQL_T_INT64))) {
        return unsup("WITHIN RECORD sum over a literal that is not an integer");
      }
      w.lit = arg->lit_bits;
    } else if (arg) {
      return unsup("WITHIN RECORD aggregates over expressions are not lowered");
    }
    q->wr_aggs.push_back(w);
    ColAccess v;
    v.name = "$" + std::to_string(i);
    v.stype = lp.return_type;
    v.mode = ColAccess::SOA;
    kp.cols.push_back(v);
    auto in = std::make_shared<Expr>();
    in->kind = Expr::INPUT;
    in->type = lp.return_type;
    in->input = i;
    scan_out.push_back(in);
    // result emission evaluates the scan select list over the group's first row:
    // above a record scan that is the per-record value itself
    lp.call = in;
    lp.is_aggregate = false;
  }
  for (uint32_t i = 0; i < plan->n_scan_select && !within; ++i) {
    err = lower_program(plan->scan_select[i], &q->scan_select[i], &u);
    if (!err.empty()) return u ? unsup(err) : Status::error(EVQL_EARG, err);
    if (q->scan_select[i].is_aggregate) return unsup("aggregate in the scan select list");
    if (!strings_lowerable(q->scan_select[i].call)) return unsup("string expression is not lowerable");
    scan_out.push_back(q->scan_select[i].call);
  }
  q->group.resize(plan->n_group);
  for (uint32_t i = 0; i < plan->n_group; ++i) {
    err = lower_program(plan->group_exprs[i], &q->group[i], &u);
    if (!err.empty()) return u ? unsup(err) : Status::error(EVQL_EARG, err);
    if (q->group[i].is_aggregate) return Status::error(EVQL_EARG, "aggregate in GROUP BY");
    ExprPtr g = inline_inputs(q->group[i].call, scan_out, &err);
    if (!err.empty()) return Status::error(EVQL_EARG, err);
    if (!strings_lowerable(g)) return unsup("string expression is not lowerable");
    mark_string_bytes(g, &kp.cols);
    kp.group.push_back(g);
  }
  if (plan->n_group == 0) {
    kp.key_mode = KEY_NONE;
  } else if (plan->n_group == 1 && kp.group[0]->type != EVQL_T_STRING &&
             kp.group[0]->type != EVQL_T_NIL) {
    kp.key_mode = KEY_EXACT;
  } else {
    kp.key_mode = KEY_HASHED;
  }

  q->select.resize(plan->n_select);
  q->select_agg_index.assign(plan->n_select, -1);
  q->select_passthrough.assign(plan->n_select, false);
  kp.need_first_row = false;
  // does anything but the group key itself read the group's first row?  (a string key that
  // is only grouped by and selected can be coded through the table's dictionary)
  bool first_row_beyond_key = false;
  for (uint32_t i = 0; i < plan->n_select; ++i) {
    LoweredProgram& lp = q->select[i];
    err = lower_program(plan->select_exprs[i], &lp, &u);
    if (!err.empty()) return u ? unsup(err) : Status::error(EVQL_EARG, err);
    if (lp.is_aggregate) {
      AggPlan a;
      a.fn = lp.aggregate_fn;
      if (a.fn != EVQL_AGG_COUNT) {
        if (lp.acc_args.size() != 1) return Status::error(EVQL_EARG, "aggregate arity");
        a.arg = inline_inputs(lp.acc_args[0], scan_out, &err);
        if (!err.empty()) return Status::error(EVQL_EARG, err);
        if (!strings_lowerable(a.arg, false)) return unsup("string aggregates are not lowered");
        mark_string_bytes(a.arg, &kp.cols);
      } else if (lp.acc_args.size() == 1 && lp.acc_args[0]->kind == Expr::CALL &&
                 lp.acc_args[0]->family == EVQL_FAM_TO_NIL) {
        // count(x): evaluate x for its side effects (division by zero) only when
        // it is not a plain column / literal
        const ExprPtr& inner = lp.acc_args[0]->args[0];
        if (inner->kind == Expr::CALL || inner->kind == Expr::IF) {
          if (!strings_lowerable(inner, false)) return unsup("string expression is not lowerable");
        }
      }
      a.first_word = int(kp.states.size());
      switch (a.fn) {
        case EVQL_AGG_COUNT:
        case EVQL_AGG_SUM_UINT64:
        case EVQL_AGG_SUM_INT64:
          kp.states.push_back({0});
          a.nwords = 1;
          break;
        case EVQL_AGG_SUM_FLOAT64:
          if (plan->float_sum_mode == EVQL_FLOAT_SUM_EXACT) {
            if (kp.n_exact >= kMaxExactSums) return unsup("too many exact float sums");
            a.exact_index = kp.n_exact++;
            kp.states.push_back({0});  // high part (signed, two's complement adds)
            kp.states.push_back({0});  // low 31 bits
            a.nwords = 2;
          } else {
            kp.states.push_back({1});
            a.nwords = 1;
          }
          break;
        case EVQL_AGG_COUNT_DISTINCT_UINT64:
          // aggregate.cc:77-137 (std::set per group): the state word counts the
          // (group, value) pairs first inserted into the aggregate's HBM pair set.
          // PartialGroupBy rows carry the set itself (sorted values, aggregate.cc:111-117):
          // read back from the pair set at emission.  Between GPUs the sets do not travel.
          if (kp.n_distinct >= kMaxDistinct) return unsup("too many count_distinct aggregates");
          a.distinct_index = kp.n_distinct++;
          kp.states.push_back({0});
          a.nwords = 1;
          break;
        case EVQL_AGG_MIN_UINT64:
        case EVQL_AGG_MAX_UINT64:
        case EVQL_AGG_MIN_INT64:
        case EVQL_AGG_MAX_INT64:
        case EVQL_AGG_MIN_FLOAT64:
        case EVQL_AGG_MAX_FLOAT64:
          kp.states.push_back({op_for_minmax(a.fn)});
          kp.states.push_back({0});
          a.nwords = 2;
          break;
        case EVQL_AGG_MEAN_UINT64:
        case EVQL_AGG_MEAN_INT64:
        case EVQL_AGG_MEAN_FLOAT64:
          kp.states.push_back({1});
          kp.states.push_back({0});
          a.nwords = 2;
          break;
        default:
          return unsup("aggregate function not lowerable");
      }
      q->select_agg_index[i] = int(kp.aggs.size());
      kp.aggs.push_back(a);
      // post-aggregate arithmetic that also reads group-level inputs needs the
      // group's first row
      if (has_input(lp.call)) {
        kp.need_first_row = true;
        first_row_beyond_key = true;
      }
    } else {
      if (has_agg_get(lp.call)) return Status::error(EVQL_EARG, "malformed aggregate program");
      ExprPtr e = inline_inputs(lp.call, scan_out, &err);
      if (!err.empty()) return Status::error(EVQL_EARG, err);
      // (no strings_lowerable check: a select expression never reaches the kernel -- it
      // is evaluated once per group at emission, over the group's first row, like the
      // reference's method_call run in GroupByExpression::nextBatch; string-producing
      // functions are fine here)
      if (kp.key_mode == KEY_EXACT && expr_equal(e, kp.group[0])) {
        q->select_passthrough[i] = true;  // value == the group key itself
      } else if (!has_input(e)) {
        // constant expression: no row needed
      } else {
        kp.need_first_row = true;
        if (!(kp.group.size() == 1 && expr_equal(e, kp.group[0]))) first_row_beyond_key = true;
      }
    }
  }
  if (int(kp.states.size()) > kMaxStateWords) return unsup("too many aggregate state words");
  // update words that travel in a partition tuple (a count's "+1" does not)
  q->n_update_words = 0;
  for (const auto& a : kp.aggs) q->n_update_words += a.fn == EVQL_AGG_COUNT ? 0 : a.nwords;
  // PartialGroupBy rows carry SHA1(tuple bytes): with a hashed identity the key
  // values have to be re-read from the group's first row
  if (plan->group_mode == EVQL_MODE_PARTIAL && kp.key_mode == KEY_HASHED) {
    kp.need_first_row = true;
  }

  kp.has_row_filter = plan->row_filter_bits != nullptr;

  // One STRING key that is only grouped by (and selected as it is): the runtime may run
  // the plan on the column's dictionary codes instead of its hashes (string_dict.cc,
  // query_prepare).  Not when WHERE or an aggregate reads the column too (they need the
  // bytes / hashes anyway), nor with count_distinct (its pair sets are keyed by the
  // identity words).
  q->dict_candidate = -1;
  if (!nested && kp.key_mode == KEY_HASHED && kp.group.size() == 1 && kp.n_distinct == 0 &&
      !first_row_beyond_key && kp.group[0]->kind == Expr::INPUT &&
      kp.group[0]->input < kp.cols.size() && kp.cols[kp.group[0]->input].string_hash) {
    const uint32_t ki = kp.group[0]->input;
    std::vector<uint32_t> used;
    if (kp.where) expr_inputs(kp.where, &used);
    for (const auto& a : kp.aggs) {
      if (a.arg) expr_inputs(a.arg, &used);
    }
    bool elsewhere = false;
    for (uint32_t u2 : used) elsewhere = elsewhere || u2 == ki;
    if (!elsewhere) q->dict_candidate = int(ki);
  }

  choose_launch_shape(&kp, plan->groups_hint);
  return Status();
}

}  // namespace evql
