/* gpu_group_by_scan.cc -- see gpu_group_by_scan.h */
#include "gpu_group_by_scan.h"
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
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
namespace {
/* a partition's tables and row filters in HBM; released when the last operator over it
 * and the registry have let go */
struct ChainResident {
  std::vector<std::shared_ptr<evql_table_t>> tables; /* scan order: newest first */
  evql_lsm_chain_t* chain;
  ChainResident() : chain(nullptr) {}
  ~ChainResident() {
    if (chain) evql_lsm_chain_destroy(chain);
  }
};
}  // namespace

GpuTableRegistry::GpuTableRegistry(int device_ordinal, uint64_t hbm_budget_bytes)
    : device_(device_ordinal), budget_(hbm_budget_bytes), ctx_(nullptr), ctx_failed_(false),
      clock_(0) {}

GpuTableRegistry::~GpuTableRegistry() {
  tables_.clear(); /* chains before the files they hold, files before the context */
  files_.clear();
  if (ctx_) evql_ctx_destroy(ctx_);
}

void GpuTableRegistry::registerTable(const std::string& table_name,
                                     const std::string& cstable_file, ScanKind kind,
                                     const std::string& version_tag) {
  /* one immutable file: no skiplist, the oldest of its chain => PartitionCursor would
   * not filter it either (partition_cursor.cc:149-151) */
  registerChain(table_name, std::vector<ChainFile>{ChainFile{cstable_file, false, false}}, kind,
                version_tag);
}

void GpuTableRegistry::registerChain(const std::string& table_name,
                                     const std::vector<ChainFile>& oldest_first, ScanKind kind,
                                     const std::string& version_tag) {
  std::unique_lock<std::mutex> lk(mutex_);
  ChainEntry e;
  e.files = oldest_first;
  e.kind = kind;
  e.version_tag = version_tag;
  tables_[table_name] = e; /* (a resident chain of the old registration is dropped) */
}

void GpuTableRegistry::unregisterTable(const std::string& table_name) {
  std::unique_lock<std::mutex> lk(mutex_);
  tables_.erase(table_name);
}

void GpuTableRegistry::setResolver(Resolver r) {
  std::unique_lock<std::mutex> lk(mutex_);
  resolver_ = r;
}

std::string GpuTableRegistry::lastError() {
  std::unique_lock<std::mutex> lk(mutex_);
  return last_error_;
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

uint64_t GpuTableRegistry::residentBytes() {
  std::unique_lock<std::mutex> lk(mutex_);
  uint64_t b = 0;
  for (const auto& f : files_) b += f.second.size;
  return b;
}

/* (mutex held) the resident table of a file; re-read when the file changed on disk */
std::shared_ptr<evql_table_t> GpuTableRegistry::openFile(const std::string& path,
                                                         std::string* signature) {
  struct stat st;
  if (stat(path.c_str(), &st) != 0) {
    last_error_ = "cannot stat " + path;
    files_.erase(path);
    return nullptr;
  }
  const int64_t mtime_ns = (int64_t) st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
  char sig[96];
  snprintf(sig, sizeof(sig), ":%llu:%lld;", (unsigned long long) st.st_size, (long long) mtime_ns);
  *signature += path + sig;
  auto it = files_.find(path);
  if (it != files_.end() &&
      (it->second.size != (uint64_t) st.st_size || it->second.mtime_ns != mtime_ns)) {
    files_.erase(it); /* the file behind the name changed: the HBM copy is stale */
    it = files_.end();
  }
  if (it == files_.end()) {
    evql_table_t* t = nullptr;
    if (evql_table_open_file(ctx_, path.c_str(), &t) != EVQL_OK) {
      last_error_ = evql_last_error();
      return nullptr;
    }
    FileEntry fe;
    fe.table = std::shared_ptr<evql_table_t>(t, evql_table_close);
    fe.size = (uint64_t) st.st_size;
    fe.mtime_ns = mtime_ns;
    it = files_.insert(std::make_pair(path, fe)).first;
  }
  it->second.last_use = ++clock_;
  return it->second.table;
}

/* (mutex held) least recently used files leave the registry until the budget holds; an
 * operator that still runs over one keeps its own reference until it ends */
void GpuTableRegistry::enforceBudget() {
  for (;;) {
    uint64_t total = 0;
    for (const auto& f : files_) total += f.second.size;
    if (total <= budget_ || files_.size() <= 1) return;
    auto victim = files_.end();
    for (auto it = files_.begin(); it != files_.end(); ++it) {
      if (it->second.last_use == clock_) continue; /* the one just asked for */
      if (victim == files_.end() || it->second.last_use < victim->second.last_use) victim = it;
    }
    if (victim == files_.end()) return;
    /* chains built over the victim hold it too: drop those residents */
    for (auto& t : tables_) {
      for (const auto& cf : t.second.files) {
        if (cf.file == victim->first) {
          t.second.resident.reset();
          t.second.signature.clear();
        }
      }
    }
    files_.erase(victim);
  }
}

bool GpuTableRegistry::lookup(const std::string& table_name, ResidentSource* out) {
  evql_ctx_t* ctx = context();
  if (!ctx) return false;
  std::unique_lock<std::mutex> lk(mutex_);
  auto it = tables_.find(table_name);
  ChainEntry resolved;
  ChainEntry* e = nullptr;
  if (it != tables_.end()) {
    e = &it->second;
  } else if (resolver_) {
    Resolver r = resolver_;
    lk.unlock(); /* (the resolver may take the server's own locks) */
    const bool ok = r(table_name, &resolved.files, &resolved.kind, &resolved.version_tag);
    lk.lock();
    if (ok) {
      /* remembered under the name so that the resident chain is reused while the
       * resolver keeps giving the same files */
      auto known = tables_.find("\001resolved:" + table_name);
      if (known != tables_.end()) {
        bool same = known->second.files.size() == resolved.files.size();
        for (size_t i = 0; same && i < resolved.files.size(); ++i) {
          same = known->second.files[i].file == resolved.files[i].file &&
                 known->second.files[i].has_skiplist == resolved.files[i].has_skiplist &&
                 known->second.files[i].has_updates == resolved.files[i].has_updates;
        }
        if (!same) known->second = resolved;
        known->second.version_tag = resolved.version_tag;
        known->second.kind = resolved.kind;
        e = &known->second;
      } else {
        e = &(tables_["\001resolved:" + table_name] = resolved);
      }
    }
  }
  if (!e) {
    last_error_ = "table not registered with the GPU executor: " + table_name;
    return false;
  }
  if (e->files.empty()) {
    last_error_ = "partition without LSM files: " + table_name;
    return false;
  }
  /* files in scan order: newest first */
  std::string signature;
  std::vector<std::shared_ptr<evql_table_t>> tabs;
  for (size_t i = e->files.size(); i-- > 0;) {
    std::shared_ptr<evql_table_t> t = openFile(e->files[i].file, &signature);
    if (!t) return false;
    tabs.push_back(t);
  }
  out->kind = e->kind;
  out->version_tag = e->version_tag;
  out->table = nullptr;
  out->chain = nullptr;
  /* does any file get a row filter?  Not when no file has a skiplist and none but the
   * oldest has updates (partition_cursor.cc:149-155 with an empty id set) */
  bool filters = false;
  for (size_t i = 0; i < e->files.size(); ++i) {
    if (e->files[i].has_skiplist || (i > 0 && e->files[i].has_updates)) filters = true;
  }
  if (e->files.size() == 1 && !filters) {
    out->table = tabs[0].get();
    out->keepalive = tabs[0];
    enforceBudget();
    return true;
  }
  if (!e->resident || e->signature != signature) {
    std::shared_ptr<ChainResident> cr(new ChainResident());
    cr->tables = tabs;
    if (evql_lsm_chain_create(ctx, &cr->chain) != EVQL_OK) {
      last_error_ = evql_last_error();
      return false;
    }
    for (size_t k = 0; k < tabs.size(); ++k) {
      const ChainFile& cf = e->files[e->files.size() - 1 - k];
      const uint32_t flags = (cf.has_skiplist ? EVQL_LSM_HAS_SKIPLIST : 0u) |
                             (cf.has_updates ? EVQL_LSM_HAS_UPDATES : 0u);
      if (evql_lsm_chain_add(cr->chain, tabs[k].get(), flags, nullptr, 0) != EVQL_OK) {
        last_error_ = evql_last_error();
        return false;
      }
    }
    /* PartitionCursor::openNextTable's filter loops, all files at once, on the device */
    if (evql_lsm_chain_build(cr->chain) != EVQL_OK) {
      last_error_ = evql_last_error();
      return false;
    }
    e->resident = cr;
    e->signature = signature;
  }
  ChainResident* cr = static_cast<ChainResident*>(e->resident.get());
  out->chain = cr->chain;
  out->keepalive = e->resident;
  enforceBudget();
  return true;
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
                               csql::ExecutionContext* execution_context, evql_query_t* query,
                               std::shared_ptr<void> source_keepalive)
    : txn_(txn), execution_context_(execution_context), query_(query),
      source_(source_keepalive), completed_(false),
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
GpuLowering::GpuLowering(std::shared_ptr<GpuTableRegistry> tables, GpuSchedulerOptions opts)
    : tables_(tables), opts_(opts), cache_hits_(new std::atomic<uint64_t>(0)) {}

csql::TableExpression* GpuLowering::tryLower(csql::Transaction* txn,
                                             csql::ExecutionContext* execution_context,
                                             csql::GroupByNode* group,
                                             csql::SequentialScanNode* seqscan,
                                             std::string* why) {
  ResidentSource src;
  if (!tables_->lookup(seqscan->tableName(), &src)) {
    *why = tables_->lastError();
    return nullptr;
  }
  PlanBuffers pb;
  // a data node serving EVQL_OP_QUERY_PARTIALAGGR gets a GroupByNode marked partial
  // (server/sql/scheduler.cc:59-64): the operator then emits PartialGroupBy rows
  const bool partial = group != nullptr && (opts_.partial || group->isPartialAggregation());
  if (!buildPlanDesc(txn, group, seqscan, src.kind, partial, &pb, why)) {
    return nullptr;
  }
  evql_query_t* q = nullptr;
  // a partition's file chain: PartitionCursor under the GROUP BY (server/sql/
  // partition_cursor.cc) -> one operator over all files, the row filters from the chain
  int rc = src.chain ? evql_query_create_chain(tables_->context(), src.chain, &pb.desc, &q)
                     : evql_query_create(tables_->context(), src.table, &pb.desc, &q);
  if (rc == EVQL_ENOTSUP) {
    *why = std::string("ENOTSUP: ") + evql_last_error();
    return nullptr;
  }
  if (rc != EVQL_OK) {
    /* a malformed plan is an error in the reference too (e.g. EARG illegal
     * column type, CSTableScan.cc:783-784) */
    RAISE(kRuntimeError, evql_last_error());
  }
  auto op = new GpuGroupByScan(txn, execution_context, q, src.keepalive);
  const std::string& version_tag = src.version_tag;
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

csql::TableExpression* GpuLowering::lowerGroupBy(csql::Transaction* txn,
                                                 csql::ExecutionContext* execution_context,
                                                 csql::GroupByNode* node) {
  std::string why;
  csql::TableExpression* op = nullptr;
  auto seqscan = dynamic_cast<csql::SequentialScanNode*>(node->inputTable().get());
  if (seqscan) {
    op = tryLower(txn, execution_context, node, seqscan, &why);
  } else {
    why = "input of the GROUP BY is not a sequential scan";
  }
  decisions_.push_back(Decision{"groupby", op != nullptr, why});
  if (!op && opts_.strict) RAISEF(kRuntimeError, "GPU executor: not lowered: $0", why);
  return op;
}

csql::TableExpression* GpuLowering::lowerScan(csql::Transaction* txn,
                                              csql::ExecutionContext* execution_context,
                                              csql::SequentialScanNode* node) {
  std::string why;
  csql::TableExpression* op = tryLower(txn, execution_context, nullptr, node, &why);
  decisions_.push_back(Decision{"seqscan", op != nullptr, why});
  if (!op && opts_.strict) RAISEF(kRuntimeError, "GPU executor: not lowered: $0", why);
  return op;
}

/* ORDER BY / LIMIT directly above a lowered GROUP BY: into the operator (its top-k runs
 * on the device over the dense group records); anything else: the reference's operators
 * stacked on the input that was just built (scheduler.cc:36-49, 95-132) */
ScopedPtr<csql::TableExpression> GpuLowering::orderBy(csql::Transaction* txn,
                                                      csql::ExecutionContext* execution_context,
                                                      csql::OrderByNode* node,
                                                      ScopedPtr<csql::TableExpression> input) {
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

ScopedPtr<csql::TableExpression> GpuLowering::limit(csql::ExecutionContext* execution_context,
                                                    csql::LimitNode* node,
                                                    ScopedPtr<csql::TableExpression> input) {
  std::string why = "input is not the GPU operator";
  auto gpu = dynamic_cast<GpuGroupByScan*>(input.get());
  const bool fused = gpu && gpu->setLimit(node->limit(), node->offset(), &why);
  decisions_.push_back(Decision{"limit", fused, fused ? "" : why});
  if (fused) return input;
  return mkScoped(new csql::LimitExpression(execution_context, node->limit(), node->offset(),
                                            std::move(input)));
}

}  // namespace evql_adapter
