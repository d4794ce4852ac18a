/*
 * gpu_group_by_scan.h -- the fused MI355X operator behind the reference's own
 * operator interface, and the scheduler hook that builds it.
 *
 * REFERENCE-SIDE ADAPTER (see gpu_bridge.h): includes reference headers, compiled
 * only inside a reference build.  In the reference tree this would live next to
 * sql/statements/select/groupby.h; it is installed exactly like
 * eventql::Scheduler (server/sql/scheduler.cc:55-77):
 *
 *     runtime->setScheduler(mkScoped(
 *         new evql_adapter::GpuSchedulerT<eventql::Scheduler>(tables, opts, config, pmap, cdir, auth)));
 *
 * GpuSchedulerT<Base> is a mix-in over any csql::DefaultScheduler subclass: it tries the
 * GPU operator first and delegates everything it does not lower to Base -- over
 * eventql::Scheduler (server/sql/scheduler.h:37) that is the partition fan-out
 * (buildPipelineGroupByExpression) and the CPU PartialGroupByExpression.
 *
 *   csql::TableExpression               sql/table_expression.h:35-50
 *   csql::DefaultScheduler              sql/scheduler.h:78-173 (virtual build*)
 *   GroupByExpression (replaced)        sql/statements/select/groupby.cc:40-229
 *   PartialGroupByExpression (replaced) sql/statements/select/groupby.cc:231-491
 *   FastCSTableScan / CSTableScan (replaced) sql/CSTableScan.cc:187-541, 688-1009
 */
#pragma once
#include <atomic>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include <eventql/sql/qtree/LimitNode.h>
#include <eventql/sql/qtree/OrderByNode.h>
#include <eventql/sql/scheduler.h>
#include <eventql/sql/table_expression.h>
#include <eventql/sql/transaction.h>
#include <eventql/util/SHA1.h>
#include <eventql/util/option.h>
#include "evql_gpu.h"
#include "gpu_bridge.h"

namespace evql_adapter {

/* which reference scan operator the table's provider would have built */
enum class ScanKind {
  FAST,   /* FastCSTableScan (CSTableScanProvider.cc:38-54, partition_cursor.cc:199) */
  DREMEL  /* CSTableScan     (partition_cursor.cc:206-213)                          */
};

/* one LSM file of a partition (db/partition_state.proto LSMTableRef) */
struct ChainFile {
  std::string file;   /* <base_path>/<filename>.cst */
  bool has_skiplist;  /* LSMTableRef::has_skiplist */
  bool has_updates;   /* LSMTableRef::has_updates  */
};

/* what a lowered operator scans; keeps the HBM-resident tables alive while it runs */
struct ResidentSource {
  evql_table_t* table = nullptr;      /* a single file without row filter, or      */
  evql_lsm_chain_t* chain = nullptr;  /* the file chain of a partition, filters built */
  ScanKind kind = ScanKind::FAST;
  std::string version_tag;
  std::shared_ptr<void> keepalive;
};

/*
 * cstable files resident in HBM, keyed by SQL table name.  The reference's
 * TableProvider hides the file name (CSTableScanProvider::cstable_file_ is
 * protected), so whoever registers the provider registers the file here too; for
 * evqld's partitions (server/sql/table_scan.cc:116-144 -> PartitionCursor) a resolver
 * maps the scan's table name ("tsdb://localhost/<table>/<partition>") to the
 * partition's file chain -- gpu_partition.h builds one from a PartitionSnapshot.
 *
 * Files are opened once and stay in HBM across queries.  Every lookup stats the file:
 * when size or mtime changed the resident copy is dropped and the file is read again.
 * Resident bytes are bounded by a budget (least recently used files go first; a table
 * still in use by a running operator is released when that operator ends).
 */
class GpuTableRegistry {
public:
  explicit GpuTableRegistry(int device_ordinal = 0, uint64_t hbm_budget_bytes = 200ull << 30);
  ~GpuTableRegistry();
  GpuTableRegistry(const GpuTableRegistry&) = delete;

  /* version_tag: what identifies this file's contents to the query cache -- the
   * "$namespace~$table~$partition~$lsm_sequence" ingredients of TableScan's cache key
   * (server/sql/table_provider.cc:229-238).  Empty: results are never cached. */
  void registerTable(const std::string& table_name, const std::string& cstable_file,
                     ScanKind kind = ScanKind::FAST, const std::string& version_tag = "");

  /* a partition: its LSM files OLDEST first, as PartitionState::lsm_tables lists them
   * (PartitionCursor walks them backwards, partition_cursor.cc:134-143) */
  void registerChain(const std::string& table_name, const std::vector<ChainFile>& oldest_first,
                     ScanKind kind = ScanKind::FAST, const std::string& version_tag = "");
  void unregisterTable(const std::string& table_name);

  /* asked for names nobody registered; false => not a GPU table (CPU operators).
   * Called on every lookup of such a name: the answer follows the partition's
   * current snapshot. */
  typedef std::function<bool(const std::string& table_name, std::vector<ChainFile>* oldest_first,
                             ScanKind* kind, std::string* version_tag)> Resolver;
  void setResolver(Resolver r);

  /* false when the name is unknown, a file cannot be opened or no device is present
   * (lastError() says which) -- the caller then keeps the CPU operators */
  bool lookup(const std::string& table_name, ResidentSource* out);
  evql_ctx_t* context();
  std::string lastError();
  uint64_t residentBytes();

private:
  struct FileEntry {
    std::shared_ptr<evql_table_t> table;
    uint64_t size = 0;
    int64_t mtime_ns = 0;
    uint64_t last_use = 0;
  };
  struct ChainEntry {
    std::vector<ChainFile> files; /* oldest first */
    ScanKind kind = ScanKind::FAST;
    std::string version_tag;
    /* built filters, valid while the files' identities are unchanged */
    std::shared_ptr<void> resident; /* ChainResident */
    std::string signature;
  };
  std::shared_ptr<evql_table_t> openFile(const std::string& path, std::string* signature);
  void enforceBudget();

  std::mutex mutex_;
  int device_;
  uint64_t budget_;
  evql_ctx_t* ctx_;
  bool ctx_failed_;
  uint64_t clock_;
  std::map<std::string, ChainEntry> tables_;
  std::map<std::string, FileEntry> files_;
  Resolver resolver_;
  std::string last_error_;
};

/* everything evql_query_create reads; owns the lowered programs */
struct PlanBuffers {
  std::vector<std::string> scan_column_names;
  std::vector<const char*> scan_column_ptrs;
  std::vector<uint32_t> scan_column_types;
  std::unique_ptr<LoweredProgram> where;
  std::vector<std::unique_ptr<LoweredProgram>> scan_select, group, select;
  std::vector<evql_program_t> scan_select_c, group_c, select_c;
  evql_plan_desc_t desc;
};

/* Fills `out` from the same objects the reference compiles its operators from
 * (scheduler.cc:153-182, CSTableScan.cc:726-755).  group == nullptr: a bare scan.
 * false => not lowerable (*why says what) */
bool buildPlanDesc(csql::Transaction* txn, csql::GroupByNode* group,
                   csql::SequentialScanNode* seqscan, ScanKind kind, bool partial,
                   PlanBuffers* out, std::string* why);

class GpuGroupByScan : public csql::TableExpression {
public:
  static const size_t kOutputBatchSize = 1024; /* groupby.h:36, CSTableScan.h:46 */

  GpuGroupByScan(csql::Transaction* txn, csql::ExecutionContext* execution_context,
                 evql_query_t* query, std::shared_ptr<void> source_keepalive = nullptr);
  ~GpuGroupByScan() override;

  ReturnCode execute() override;
  ReturnCode nextBatch(csql::SVector* columns, size_t* len) override;
  size_t getColumnCount() const override;
  csql::SType getColumnType(size_t idx) const override;

  /* PartialGroupByExpression's twin keeps its rows in the runtime's QueryCache under
   * this key, like the operator it replaces (groupby.cc:255-296, 410-432, 474-482);
   * hits are counted in *hit_counter when given */
  void enableQueryCache(const SHA1Hash& key, std::shared_ptr<std::atomic<uint64_t>> hit_counter);
  Option<SHA1Hash> getCacheKey() const override;

  /* ORDER BY .. LIMIT fused into the operator (OrderByExpression::execute,
   * orderby.cc:60-160; LimitExpression::nextBatch, limit.cc:52-125): the sort
   * expressions are programs over this operator's output columns.  false => not
   * fusable (*why says what); the caller then stacks the CPU operators on top */
  void markPartial() { partial_ = true; }
  bool setOrder(const std::vector<csql::OrderByNode::SortSpec>& specs, std::string* why);
  bool setLimit(size_t limit, size_t offset, std::string* why);

private:
  static int heartbeat(void* self);
  static const size_t kMaxCachedBytes = 256u << 20;
  struct CachedBatch {
    size_t nrows;
    std::vector<std::string> columns; /* packed SVector bytes */
  };
  csql::Transaction* txn_;
  csql::ExecutionContext* execution_context_;
  evql_query_t* query_;
  std::shared_ptr<void> source_; /* the tables / filter chain the query reads */
  bool completed_;
  bool partial_ = false;
  Option<SHA1Hash> cache_key_;
  std::shared_ptr<std::atomic<uint64_t>> cache_hits_;
  bool from_cache_;
  bool recording_;
  size_t recorded_bytes_;
  size_t replay_pos_;
  std::vector<CachedBatch> batches_;
  std::vector<std::unique_ptr<LoweredProgram>> sort_programs_;
  std::vector<evql_sort_spec_t> sort_specs_;
};

struct GpuSchedulerOptions {
  GpuSchedulerOptions() : lower_group_by(true), lower_scans(false), partial(false),
                          strict(false), fuse_order_by(true) {}
  bool lower_group_by; /* GroupByExpression + scan -> one fused operator */
  bool fuse_order_by;  /* ORDER BY / LIMIT directly above it -> into the operator */
  bool lower_scans;    /* bare FastCSTableScan / CSTableScan -> GPU scan operator */
  bool partial;        /* build PartialGroupByExpression's twin (a data node) */
  bool strict;         /* tests: RAISE instead of falling back to the CPU operators */
};

/* the part of the scheduler hook that does not depend on the base scheduler */
class GpuLowering {
public:
  GpuLowering(std::shared_ptr<GpuTableRegistry> tables, GpuSchedulerOptions opts);

  struct Decision {
    std::string node;     /* "groupby" | "seqscan" | "orderby" | "limit" */
    bool lowered;
    std::string reason;   /* why not, when !lowered */
  };

  /* nullptr => keep the CPU operators (*why says why) */
  csql::TableExpression* tryLower(csql::Transaction* txn,
                                  csql::ExecutionContext* execution_context,
                                  csql::GroupByNode* group, csql::SequentialScanNode* seqscan,
                                  std::string* why);
  /* GROUP BY node -> operator, or nullptr; records the decision; RAISEs in strict mode */
  csql::TableExpression* lowerGroupBy(csql::Transaction* txn, csql::ExecutionContext* ctx,
                                      csql::GroupByNode* node);
  csql::TableExpression* lowerScan(csql::Transaction* txn, csql::ExecutionContext* ctx,
                                   csql::SequentialScanNode* node);
  /* ORDER BY / LIMIT above an operator that was just built: fused into it when it is the
   * GPU operator, the reference's CPU operator stacked on it otherwise
   * (sql/scheduler.cc:36-49, 95-132) */
  ScopedPtr<csql::TableExpression> orderBy(csql::Transaction* txn, csql::ExecutionContext* ctx,
                                           csql::OrderByNode* node,
                                           ScopedPtr<csql::TableExpression> input);
  ScopedPtr<csql::TableExpression> limit(csql::ExecutionContext* ctx, csql::LimitNode* node,
                                         ScopedPtr<csql::TableExpression> input);

  std::shared_ptr<GpuTableRegistry> tables_;
  GpuSchedulerOptions opts_;
  std::vector<Decision> decisions_;
  std::shared_ptr<std::atomic<uint64_t>> cache_hits_;
};

/*
 * The scheduler hook as a mix-in over any csql::DefaultScheduler subclass.
 *   GpuSchedulerT<csql::DefaultScheduler>   standalone csql (tests, the probe)
 *   GpuSchedulerT<eventql::Scheduler>       evqld: the object handed to
 *                                           Runtime::setScheduler INSTEAD of
 *                                           eventql::Scheduler; what is not lowered
 *                                           reaches eventql::Scheduler's overrides
 *                                           unchanged (server/sql/scheduler.cc:55-162)
 * Base's constructor arguments follow (tables, opts).
 */
template <class Base>
class GpuSchedulerT : public Base {
public:
  template <class... BaseArgs>
  GpuSchedulerT(std::shared_ptr<GpuTableRegistry> tables, GpuSchedulerOptions opts,
                BaseArgs&&... base_args)
      : Base(std::forward<BaseArgs>(base_args)...), gpu_(tables, opts) {}

  /* diagnostics: how the last build* calls were answered */
  typedef GpuLowering::Decision Decision;
  const std::vector<Decision>& decisions() const { return gpu_.decisions_; }
  void clearDecisions() { gpu_.decisions_.clear(); }
  uint64_t queryCacheHits() const { return gpu_.cache_hits_->load(); }

protected:
  ScopedPtr<csql::TableExpression> buildGroupByExpression(
      csql::Transaction* txn, csql::ExecutionContext* execution_context,
      RefPtr<csql::GroupByNode> node) override {
    if (gpu_.opts_.lower_group_by) {
      csql::TableExpression* op = gpu_.lowerGroupBy(txn, execution_context, node.get());
      if (op) return ScopedPtr<csql::TableExpression>(op);
    }
    return Base::buildGroupByExpression(txn, execution_context, node);
  }

  ScopedPtr<csql::TableExpression> buildSequentialScan(
      csql::Transaction* txn, csql::ExecutionContext* execution_context,
      RefPtr<csql::SequentialScanNode> node) override {
    if (gpu_.opts_.lower_scans) {
      csql::TableExpression* op = gpu_.lowerScan(txn, execution_context, node.get());
      if (op) return ScopedPtr<csql::TableExpression>(op);
    }
    return Base::buildSequentialScan(txn, execution_context, node);
  }

  ScopedPtr<csql::TableExpression> buildOrderByExpression(
      csql::Transaction* txn, csql::ExecutionContext* execution_context,
      RefPtr<csql::OrderByNode> node) override {
    if (!gpu_.opts_.lower_group_by || !gpu_.opts_.fuse_order_by) {
      return Base::buildOrderByExpression(txn, execution_context, node);
    }
    return gpu_.orderBy(txn, execution_context, node.get(),
                        this->buildTableExpression(
                            txn, execution_context,
                            node->inputTable().template asInstanceOf<csql::TableExpressionNode>()));
  }

  ScopedPtr<csql::TableExpression> buildLimit(
      csql::Transaction* txn, csql::ExecutionContext* execution_context,
      RefPtr<csql::LimitNode> node) override {
    if (!gpu_.opts_.lower_group_by || !gpu_.opts_.fuse_order_by) {
      return Base::buildLimit(txn, execution_context, node);
    }
    return gpu_.limit(execution_context, node.get(),
                      this->buildTableExpression(
                          txn, execution_context,
                          node->inputTable().template asInstanceOf<csql::TableExpressionNode>()));
  }

  GpuLowering gpu_;
};

typedef GpuSchedulerT<csql::DefaultScheduler> GpuScheduler;

const char* statusCodeString(int evql_status_code);

}  // namespace evql_adapter
