/* gpu_group_by_scan.cc -- see gpu_group_by_scan.h */
#include "gpu_group_by_scan.h"
#include <string.h>
#include <eventql/sql/qtree/GroupByNode.h>
#include <eventql/sql/qtree/SequentialScanNode.h>
#include <eventql/sql/runtime/QueryBuilder.h>
#include <eventql/sql/runtime/ValueExpression.h>
#include <eventql/sql/runtime/query_cache.h>
#include <eventql/sql/runtime/runtime.h>
#include <eventql/sql/statements/select/limit.h>
#include <eventql/sql/statements/select/orderby.h>
#include <eventql/util/exception.h>
#include <eventql/util/io/inputstream.h>
#include <eventql/util/io/outputstream.h>
#include <eventql/util/stringutil.h>

namespace evql_adapter {

const char* statusCodeString(int rc) {
  switch (rc) {
    case EVQL_EIO: return "EIO";
    case EVQL_EARG: return "EARG";
    default: return "ERUNTIME"; /* util/return_code.h:32-80 knows no other codes */
  }
}

/* ---------------------------------------------------------------- registry */
GpuTableRegistry::GpuTableRegistry(int device_ordinal)
    : device_(device_ordinal), ctx_(nullptr), ctx_failed_(false) {}

GpuTableRegistry::~GpuTableRegistry() {
  for (auto& t : tables_) {
    if (t.second.table) evql_table_close(t.second.table);
  }
  if (ctx_) evql_ctx_destroy(ctx_);
}

void GpuTableRegistry::registerTable(const std::string& table_name,
                                     const std::string& cstable_file, ScanKind kind,
                                     const std::string& version_tag) {
  std::unique_lock<std::mutex> lk(mutex_);
  auto it = tables_.find(table_name);
  if (it != tables_.end() && it->second.table) evql_table_close(it->second.table);
  tables_[table_name] = Entry{cstable_file, kind, nullptr, version_tag};
}

evql_ctx_t* GpuTableRegistry::context() {
  std::unique_lock<std::mutex> lk(mutex_);
  if (!ctx_ && !ctx_failed_) {
    if (evql_ctx_create(device_, nullptr, &ctx_) != EVQL_OK) {
      ctx_failed_ = true; /* no MI355X: every query keeps the CPU operators */
      ctx_ = nullptr;
      last_error_ = evql_last_error();
    }
  }
  return ctx_;
}

evql_table_t* GpuTableRegistry::lookup(const std::string& table_name, ScanKind* kind,
                                       std::string* version_tag) {
  evql_ctx_t* ctx = context();
  if (!ctx) return nullptr;
  std::unique_lock<std::mutex> lk(mutex_);
  auto it = tables_.find(table_name);
  if (it == tables_.end()) {
    last_error_ = "table not registered with the GPU executor: " + table_name;
    return nullptr;
  }
  if (!it->second.table) {
    if (evql_table_open_file(ctx, it->second.file.c_str(), &it->second.table) != EVQL_OK) {
      last_error_ = evql_last_error();
      it->second.table = nullptr;
      return nullptr;
    }
  }
  if (kind) *kind = it->second.kind;
  if (version_tag) *version_tag = it->second.version_tag;
  return it->second.table;
}

/* ------------------------------------------------------------ plan lowering */
static bool lowerExpression(csql::Transaction* txn, RefPtr<csql::ValueExpressionNode> expr,
                            std::unique_ptr<LoweredProgram>* out, std::string* why) {
  /* the same call the reference's operators make (scheduler.cc:161-170,
   * CSTableScan.cc:727-735): qtree -> vm::Program through the runtime's compiler */
  csql::ValueExpression compiled = txn->getCompiler()->buildValueExpression(txn, expr);
  out->reset(new LoweredProgram());
  return lowerProgram(compiled.program(), out->get(), why);
}

bool buildPlanDesc(csql::Transaction* txn, csql::GroupByNode* group,
                   csql::SequentialScanNode* seqscan, ScanKind kind, bool partial,
                   PlanBuffers* pb, std::string* why) {
  memset(&pb->desc, 0, sizeof(pb->desc));

  /* X_INPUT space of WHERE and the scan select list: selectedColumns() ==
   * input_columns_ (qtree/SequentialScanNode.cc:151-159), the order in which
   * FastCSTableScan::execute opens its column readers (CSTableScan.cc:743-752) */
  pb->scan_column_names = seqscan->selectedColumns();
  for (size_t i = 0; i < pb->scan_column_names.size(); ++i) {
    pb->scan_column_ptrs.push_back(pb->scan_column_names[i].c_str());
    pb->scan_column_types.push_back((uint32_t) seqscan->getInputColumnType(i));
  }

  auto where = seqscan->whereExpression();
  if (!where.isEmpty()) {
    if (!lowerExpression(txn, where.get(), &pb->where, why)) return false;
  }
  for (const auto& sl : seqscan->selectList()) {
    pb->scan_select.emplace_back();
    if (!lowerExpression(txn, sl->expression(), &pb->scan_select.back(), why)) return false;
    pb->scan_select_c.push_back(pb->scan_select.back()->c);
  }
  if (group) {
    /* X_INPUT(j) of these programs indexes the scan's select list
     * (scheduler.cc:153-182: compiled exactly like this) */
    for (const auto& e : group->groupExpressions()) {
      pb->group.emplace_back();
      if (!lowerExpression(txn, e, &pb->group.back(), why)) return false;
      pb->group_c.push_back(pb->group.back()->c);
    }
    for (const auto& sl : group->selectList()) {
      pb->select.emplace_back();
      if (!lowerExpression(txn, sl->expression(), &pb->select.back(), why)) return false;
      pb->select_c.push_back(pb->select.back()->c);
    }
  }

  evql_plan_desc_t& d = pb->desc;
  d.scan_columns = pb->scan_column_ptrs.data();
  d.scan_column_types = pb->scan_column_types.data();
  d.n_scan_columns = (uint32_t) pb->scan_column_ptrs.size();
  d.where = pb->where ? &pb->where->c : nullptr;
  d.scan_select = pb->scan_select_c.data();
  d.n_scan_select = (uint32_t) pb->scan_select_c.size();
  d.group_exprs = pb->group_c.data();
  d.n_group = (uint32_t) pb->group_c.size();
  d.select_exprs = pb->select_c.data();
  d.n_select = (uint32_t) pb->select_c.size();
  d.group_mode = partial ? EVQL_MODE_PARTIAL : EVQL_MODE_FINAL;

  /* which reference scan operator is being replaced */
  if (kind == ScanKind::FAST) {
    d.scan_mode = EVQL_SCAN_FLAT;
  } else {
    switch (seqscan->aggregationStrategy()) {
      case csql::AggregationStrategy::NO_AGGREGATION:
        d.scan_mode = EVQL_SCAN_NESTED;
        break;
      case csql::AggregationStrategy::AGGREGATE_WITHIN_RECORD_FLAT:
        d.scan_mode = EVQL_SCAN_NESTED_WITHIN_RECORD;
        break;
      default:
        if (why) *why = "aggregation strategy not lowered";
        return false;
    }
  }
  return true;
}

/* ----------------------------------------------------------------- operator */
GpuGroupByScan::GpuGroupByScan(csql::Transaction* txn,
                               csql::ExecutionContext* execution_context, evql_query_t* query)
    : txn_(txn), execution_context_(execution_context), query_(query), completed_(false),
      from_cache_(false), recording_(false), recorded_bytes_(0), replay_pos_(0) {
  execution_context_->incrementNumTasks(); /* groupby.cc:54 */
}

void GpuGroupByScan::enableQueryCache(const SHA1Hash& key,
                                      std::shared_ptr<std::atomic<uint64_t>> hit_counter) {
  cache_key_ = Some(key);
  cache_hits_ = hit_counter;
}

Option<SHA1Hash> GpuGroupByScan::getCacheKey() const { return cache_key_; }

bool GpuGroupByScan::setOrder(const std::vector<csql::OrderByNode::SortSpec>& specs,
                              std::string* why) {
  if (partial_) { /* (its rows are group keys + saved states: nothing to order by) */
    *why = "partial aggregate";
    return false;
  }
  std::vector<std::unique_ptr<LoweredProgram>> programs;
  std::vector<evql_sort_spec_t> lowered;
  for (const auto& ss : specs) {
    /* compiled exactly like scheduler.cc:101-104; X_INPUT i = output column i */
    csql::ValueExpression compiled = txn_->getCompiler()->buildValueExpression(txn_, ss.expr);
    programs.emplace_back(new LoweredProgram());
    if (!lowerProgram(compiled.program(), programs.back().get(), why)) return false;
    evql_sort_spec_t sp;
    sp.expr = programs.back()->c;
    sp.descending = ss.descending ? 1 : 0;
    lowered.push_back(sp);
  }
  int rc = evql_query_set_order(query_, lowered.data(), (uint32_t) lowered.size(), -1, 0);
  if (rc != EVQL_OK) {
    *why = std::string(rc == EVQL_ENOTSUP ? "ENOTSUP: " : "") + evql_last_error();
    return false;
  }
  sort_programs_ = std::move(programs);
  sort_specs_ = std::move(lowered);
  return true;
}

bool GpuGroupByScan::setLimit(size_t limit, size_t offset, std::string* why) {
  if (partial_) {
    *why = "partial aggregate";
    return false;
  }
  int rc = evql_query_set_order(query_, sort_specs_.data(), (uint32_t) sort_specs_.size(),
                                (int64_t) limit, (uint64_t) offset);
  if (rc != EVQL_OK) {
    *why = std::string(rc == EVQL_ENOTSUP ? "ENOTSUP: " : "") + evql_last_error();
    return false;
  }
  return true;
}

GpuGroupByScan::~GpuGroupByScan() { evql_query_destroy(query_); }

int GpuGroupByScan::heartbeat(void* self) {
  auto op = static_cast<GpuGroupByScan*>(self);
  return op->txn_->triggerHeartbeat().isSuccess() ? 0 : 1; /* groupby.cc:100-105 */
}

ReturnCode GpuGroupByScan::execute() {
  execution_context_->incrementNumTasksRunning(); /* groupby.cc:70 */
  /* read cache (groupby.cc:255-296).  The entry holds this operator's own output
   * batches -- packed SVector bytes per column -- not the CPU operator's group map:
   * the key is salted ("~mi355x", GpuScheduler::tryLower) so the two never mix. */
  csql::QueryCache* cache = cache_key_.isEmpty() ? nullptr : txn_->getRuntime()->getQueryCache();
  if (cache) {
    cache->getEntry(cache_key_.get(), [this](InputStream* is) {
      if (is->readUInt8() != 0x02) return;
      uint64_t nbatches = is->readUInt64();
      uint64_t ncols = is->readUInt64();
      if (ncols != getColumnCount()) return;
      std::vector<CachedBatch> got;
      for (uint64_t b = 0; b < nbatches; ++b) {
        CachedBatch cb;
        cb.nrows = is->readUInt64();
        for (uint64_t c = 0; c < ncols; ++c) {
          uint64_t sz = is->readUInt64();
          std::string bytes(sz, 0);
          if (sz) is->readNextBytes(&bytes[0], sz);
          cb.columns.emplace_back(std::move(bytes));
        }
        got.emplace_back(std::move(cb));
      }
      batches_ = std::move(got);
      from_cache_ = true;
    });
    if (from_cache_) {
      if (cache_hits_) cache_hits_->fetch_add(1);
      return ReturnCode::success();
    }
    recording_ = true;
  }
  int rc = evql_query_execute(query_, &GpuGroupByScan::heartbeat, this);
  if (rc != EVQL_OK) {
    return ReturnCode::error(statusCodeString(rc), evql_last_error());
  }
  return ReturnCode::success();
}

ReturnCode GpuGroupByScan::nextBatch(csql::SVector* columns, size_t* len) {
  size_t ncols = getColumnCount();
  if (from_cache_) {
    *len = 0;
    if (replay_pos_ < batches_.size()) {
      const CachedBatch& cb = batches_[replay_pos_++];
      for (size_t i = 0; i < ncols; ++i) {
        if (!cb.columns[i].empty()) columns[i].append(cb.columns[i].data(), cb.columns[i].size());
      }
      *len = cb.nrows;
    } else if (!completed_) {
      completed_ = true;
      execution_context_->incrementNumTasksCompleted();
    }
    return ReturnCode::success();
  }
  std::vector<evql_column_buf_t> bufs(ncols);
  int rc = evql_query_next_batch(query_, kOutputBatchSize, bufs.data(), len);
  if (rc != EVQL_OK) {
    return ReturnCode::error(statusCodeString(rc), evql_last_error());
  }
  for (size_t i = 0; i < ncols; ++i) {
    /* the library hands out packed SVector elements (svalue.cc:410-517) */
    if (bufs[i].size > 0) columns[i].append(bufs[i].data, bufs[i].size);
  }
  if (recording_ && *len > 0) {
    CachedBatch cb;
    cb.nrows = *len;
    for (size_t i = 0; i < ncols; ++i) {
      cb.columns.emplace_back(reinterpret_cast<const char*>(bufs[i].data), bufs[i].size);
      recorded_bytes_ += bufs[i].size;
    }
    batches_.emplace_back(std::move(cb));
    if (recorded_bytes_ > kMaxCachedBytes) { /* too large to be worth a cache file */
      recording_ = false;
      batches_.clear();
    }
  }
  if (*len == 0 && !completed_) {
    completed_ = true;
    execution_context_->incrementNumTasksCompleted(); /* groupby.cc:211 */
    /* store cache (groupby.cc:410-432) once the consumer has drained the operator */
    csql::QueryCache* cache = recording_ ? txn_->getRuntime()->getQueryCache() : nullptr;
    if (cache) {
      cache->storeEntry(cache_key_.get(), [this, ncols](OutputStream* os) {
        os->appendUInt8(0x02);
        os->appendUInt64(batches_.size());
        os->appendUInt64(ncols);
        for (const auto& cb : batches_) {
          os->appendUInt64(cb.nrows);
          for (const auto& c : cb.columns) {
            os->appendUInt64(c.size());
            if (!c.empty()) os->write(c.data(), c.size());
          }
        }
      });
    }
    recording_ = false;
    batches_.clear();
  }
  return ReturnCode::success();
}

size_t GpuGroupByScan::getColumnCount() const {
  return (size_t) evql_query_column_count(query_);
}

csql::SType GpuGroupByScan::getColumnType(size_t idx) const {
  return (csql::SType) evql_query_column_type(query_, (int) idx);
}

/* ---------------------------------------------------------------- scheduler */
GpuScheduler::GpuScheduler(std::shared_ptr<GpuTableRegistry> tables, GpuSchedulerOptions opts)
    : tables_(tables), opts_(opts), cache_hits_(new std::atomic<uint64_t>(0)) {}

csql::TableExpression* GpuScheduler::tryLower(csql::Transaction* txn,
                                              csql::ExecutionContext* execution_context,
                                              csql::GroupByNode* group,
                                              csql::SequentialScanNode* seqscan,
                                              std::string* why) {
  ScanKind kind = ScanKind::FAST;
  std::string version_tag;
  evql_table_t* table = tables_->lookup(seqscan->tableName(), &kind, &version_tag);
  if (!table) {
    *why = tables_->lastError();
    return nullptr;
  }
  PlanBuffers pb;
  // a data node serving EVQL_OP_QUERY_PARTIALAGGR gets a GroupByNode marked partial
  // (server/sql/scheduler.cc:59-64): the operator then emits PartialGroupBy rows
  const bool partial = group != nullptr && (opts_.partial || group->isPartialAggregation());
  if (!buildPlanDesc(txn, group, seqscan, kind, partial, &pb, why)) {
    return nullptr;
  }
  evql_query_t* q = nullptr;
  int rc = evql_query_create(tables_->context(), table, &pb.desc, &q);
  if (rc == EVQL_ENOTSUP) {
    *why = std::string("ENOTSUP: ") + evql_last_error();
    return nullptr;
  }
  if (rc != EVQL_OK) {
    /* a malformed plan is an error in the reference too (e.g. EARG illegal
     * column type, CSTableScan.cc:783-784) */
    RAISE(kRuntimeError, evql_last_error());
  }
  auto op = new GpuGroupByScan(txn, execution_context, q);
  if (partial) op->markPartial();
  if (partial && !version_tag.empty()) {
    /* PartialGroupByExpression::getCacheKey = SHA1(input key + fingerprint of the
     * group-by's expressions) (groupby.cc:474-482, server/sql/scheduler.cc:85-103);
     * the input's key = SHA1 over the scan's select list + WHERE and the table
     * version (server/sql/table_provider.cc:216-238) */
    SHA1Hash scan_fp;
    for (const auto& sl : seqscan->selectList()) {
      scan_fp = SHA1::compute(scan_fp.toString() + sl->toString());
    }
    if (!seqscan->whereExpression().isEmpty()) {
      scan_fp = SHA1::compute(scan_fp.toString() + seqscan->whereExpression().get()->toString());
    }
    SHA1Hash scan_key =
        SHA1::compute(StringUtil::format("$0~$1~$2", "", scan_fp.toString(), version_tag));
    SHA1Hash group_fp;
    for (const auto& sl : group->selectList()) {
      group_fp = SHA1::compute(group_fp.toString() + sl->toString());
    }
    for (const auto& e : group->groupExpressions()) {
      group_fp = SHA1::compute(group_fp.toString() + e->toString());
    }
    op->enableQueryCache(
        SHA1::compute(scan_key.toString() + group_fp.toString() + "~mi355x"), cache_hits_);
  }
  return op;
}

ScopedPtr<csql::TableExpression> GpuScheduler::buildGroupByExpression(
    csql::Transaction* txn, csql::ExecutionContext* execution_context,
    RefPtr<csql::GroupByNode> node) {
  if (opts_.lower_group_by) {
    std::string why;
    csql::TableExpression* op = nullptr;
    auto seqscan = dynamic_cast<csql::SequentialScanNode*>(node->inputTable().get());
    if (seqscan) {
      op = tryLower(txn, execution_context, node.get(), seqscan, &why);
    } else {
      why = "input of the GROUP BY is not a sequential scan";
    }
    decisions_.push_back(Decision{"groupby", op != nullptr, why});
    if (op) return ScopedPtr<csql::TableExpression>(op);
    if (opts_.strict) RAISEF(kRuntimeError, "GPU executor: not lowered: $0", why);
  }
  return csql::DefaultScheduler::buildGroupByExpression(txn, execution_context, node);
}

ScopedPtr<csql::TableExpression> GpuScheduler::buildSequentialScan(
    csql::Transaction* txn, csql::ExecutionContext* execution_context,
    RefPtr<csql::SequentialScanNode> node) {
  if (opts_.lower_scans) {
    std::string why;
    csql::TableExpression* op = tryLower(txn, execution_context, nullptr, node.get(), &why);
    decisions_.push_back(Decision{"seqscan", op != nullptr, why});
    if (op) return ScopedPtr<csql::TableExpression>(op);
    if (opts_.strict) RAISEF(kRuntimeError, "GPU executor: not lowered: $0", why);
  }
  return csql::DefaultScheduler::buildSequentialScan(txn, execution_context, node);
}

/* ORDER BY / LIMIT directly above a lowered GROUP BY: into the operator (its top-k runs
 * on the device over the dense group records); anything else: the reference's operators
 * stacked on the input that was just built (scheduler.cc:36-49, 95-132) */
ScopedPtr<csql::TableExpression> GpuScheduler::buildOrderByExpression(
    csql::Transaction* txn, csql::ExecutionContext* execution_context,
    RefPtr<csql::OrderByNode> node) {
  if (!opts_.lower_group_by || !opts_.fuse_order_by) {
    return csql::DefaultScheduler::buildOrderByExpression(txn, execution_context, node);
  }
  auto input = buildTableExpression(
      txn, execution_context, node->inputTable().asInstanceOf<csql::TableExpressionNode>());
  std::string why = "input is not the GPU operator";
  auto gpu = dynamic_cast<GpuGroupByScan*>(input.get());
  const bool fused = gpu && gpu->setOrder(node->sortSpecs(), &why);
  decisions_.push_back(Decision{"orderby", fused, fused ? "" : why});
  if (fused) return input;
  Vector<csql::OrderByExpression::SortExpr> sort_exprs;
  Vector<csql::PureSFunctionPtr> comparators;
  for (const auto& ss : node->sortSpecs()) {
    csql::OrderByExpression::SortExpr se;
    se.descending = ss.descending;
    se.expr = txn->getCompiler()->buildValueExpression(txn, ss.expr);
    const csql::SymbolTableEntry* symbol = nullptr;
    auto rc = txn->getSymbolTable()->resolve(
        "cmp", {se.expr.getReturnType(), se.expr.getReturnType()}, &symbol, false);
    if (!rc.isSuccess()) RAISE(kRuntimeError, rc.getMessage());
    comparators.emplace_back(symbol->getFunction()->vtable.call);
    sort_exprs.emplace_back(std::move(se));
  }
  return mkScoped(new csql::OrderByExpression(txn, execution_context, std::move(sort_exprs),
                                              comparators, std::move(input)));
}

ScopedPtr<csql::TableExpression> GpuScheduler::buildLimit(
    csql::Transaction* txn, csql::ExecutionContext* execution_context,
    RefPtr<csql::LimitNode> node) {
  if (!opts_.lower_group_by || !opts_.fuse_order_by) {
    return csql::DefaultScheduler::buildLimit(txn, execution_context, node);
  }
  auto input = buildTableExpression(
      txn, execution_context, node->inputTable().asInstanceOf<csql::TableExpressionNode>());
  std::string why = "input is not the GPU operator";
  auto gpu = dynamic_cast<GpuGroupByScan*>(input.get());
  const bool fused = gpu && gpu->setLimit(node->limit(), node->offset(), &why);
  decisions_.push_back(Decision{"limit", fused, fused ? "" : why});
  if (fused) return input;
  return mkScoped(new csql::LimitExpression(execution_context, node->limit(), node->offset(),
                                            std::move(input)));
}

}  // namespace evql_adapter
