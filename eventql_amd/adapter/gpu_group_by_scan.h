/*
 * gpu_group_by_scan.h -- the fused MI355X operator behind the reference's own
 * operator interface, and the scheduler hook that builds it.
 *
 * REFERENCE-SIDE ADAPTER (see gpu_bridge.h): includes reference headers, compiled
 * only inside a reference build.  In the reference tree this would live next to
 * sql/statements/select/groupby.h; it is installed exactly like
 * eventql::Scheduler (server/sql/scheduler.cc:55-77):
 *
 *     runtime->setScheduler(mkScoped(new evql_adapter::GpuScheduler(opts)));
 *
 *   csql::TableExpression               sql/table_expression.h:35-50
 *   csql::DefaultScheduler              sql/scheduler.h:78-173 (virtual build*)
 *   GroupByExpression (replaced)        sql/statements/select/groupby.cc:40-229
 *   PartialGroupByExpression (replaced) sql/statements/select/groupby.cc:231-491
 *   FastCSTableScan / CSTableScan (replaced) sql/CSTableScan.cc:187-541, 688-1009
 */
#pragma once
#include <atomic>
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

/*
 * cstable files resident in HBM, keyed by SQL table name.  The reference's
 * TableProvider hides the file name (CSTableScanProvider::cstable_file_ is
 * protected), so whoever registers the provider registers the file here too; in
 * evqld the equivalent call sits in PartitionCursor::openNextTable
 * (server/sql/partition_cursor.cc:197-213), where the file name is at hand.
 */
class GpuTableRegistry {
public:
  explicit GpuTableRegistry(int device_ordinal = 0);
  ~GpuTableRegistry();
  GpuTableRegistry(const GpuTableRegistry&) = delete;

  /* version_tag: what identifies this file's contents to the query cache -- the
   * "$namespace~$table~$partition~$lsm_sequence" ingredients of TableScan's cache key
   * (server/sql/table_provider.cc:229-238).  Empty: results are never cached. */
  void registerTable(const std::string& table_name, const std::string& cstable_file,
                     ScanKind kind = ScanKind::FAST, const std::string& version_tag = "");

  /* opens (once) and returns the resident table; nullptr when the name is unknown
   * or no device is present -- the caller then keeps the CPU operators */
  evql_table_t* lookup(const std::string& table_name, ScanKind* kind,
                       std::string* version_tag = nullptr);
  evql_ctx_t* context();
  const std::string& lastError() const { return last_error_; }

private:
  struct Entry {
    std::string file;
    ScanKind kind;
    evql_table_t* table;
    std::string version_tag;
  };
  std::mutex mutex_;
  int device_;
  evql_ctx_t* ctx_;
  bool ctx_failed_;
  std::map<std::string, Entry> tables_;
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
                 evql_query_t* query);
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

class GpuScheduler : public csql::DefaultScheduler {
public:
  GpuScheduler(std::shared_ptr<GpuTableRegistry> tables, GpuSchedulerOptions opts);

  /* diagnostics: how the last build* calls were answered */
  struct Decision {
    std::string node;     /* "groupby" | "seqscan" */
    bool lowered;
    std::string reason;   /* why not, when !lowered */
  };
  const std::vector<Decision>& decisions() const { return decisions_; }
  void clearDecisions() { decisions_.clear(); }
  uint64_t queryCacheHits() const { return cache_hits_->load(); }

protected:
  ScopedPtr<csql::TableExpression> buildGroupByExpression(
      csql::Transaction* txn, csql::ExecutionContext* execution_context,
      RefPtr<csql::GroupByNode> node) override;

  ScopedPtr<csql::TableExpression> buildSequentialScan(
      csql::Transaction* txn, csql::ExecutionContext* execution_context,
      RefPtr<csql::SequentialScanNode> node) override;

  ScopedPtr<csql::TableExpression> buildOrderByExpression(
      csql::Transaction* txn, csql::ExecutionContext* execution_context,
      RefPtr<csql::OrderByNode> node) override;

  ScopedPtr<csql::TableExpression> buildLimit(
      csql::Transaction* txn, csql::ExecutionContext* execution_context,
      RefPtr<csql::LimitNode> node) override;

  /* nullptr => keep the CPU operators */
  csql::TableExpression* tryLower(csql::Transaction* txn,
                                  csql::ExecutionContext* execution_context,
                                  csql::GroupByNode* group, csql::SequentialScanNode* seqscan,
                                  std::string* why);

  std::shared_ptr<GpuTableRegistry> tables_;
  GpuSchedulerOptions opts_;
  std::vector<Decision> decisions_;
  std::shared_ptr<std::atomic<uint64_t>> cache_hits_;
};

const char* statusCodeString(int evql_status_code);

}  // namespace evql_adapter
