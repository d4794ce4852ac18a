/*
 * oracle/ref_csql/probe.cc -- TEST INFRASTRUCTURE ONLY.
 *
 * A small driver around the REAL reference csql engine (csql::Runtime, parser,
 * planner, VM, GroupByExpression, FastCSTableScan / CSTableScan), compiled from
 * the sources under /root/reference by oracle/ref_csql/build.sh and linked with
 * the reference-side adapter (eventql_amd/adapter/) and libevql_mi355x.so.
 * It follows the pattern of the reference's own SQL test driver
 * (test/sql_tests.cc:233-274).
 *
 * Uses:
 *   * tests/golden/gen_ref_csql.py runs SQL through the unmodified CPU operators
 *     and commits the results / compiled vm::Programs / PartialGroupBy bytes as
 *     fixtures (the oracle's csql half is pinned on them);
 *   * on a GPU box the same binary runs the same SQL with `MODE gpu`: the
 *     reference's parser and planner build the query tree, GpuScheduler replaces
 *     GroupByExpression + the scan with the fused MI355X operator, ResultCursor
 *     pulls from it -- the drop-in demonstrated end to end;
 *   * bench.py's cpu_baseline leg (kind "reference") times the CPU operators.
 *
 * Protocol: commands on stdin, one per line; one JSON line per SQL on stdout.
 *   TABLE <name> <file.cst> [fast|dremel] [version tag]  register a table (provider + GPU
 *                                           registry; a tag makes partial results cacheable)
 *   CACHE <dir>                             install a csql::QueryCache (stores at first use)
 *   MODE cpu|gpu|gpuscan [partial] [strict] which operators execute
 *   MODE evqld [strict]                     the scheduler evqld would install:
 *                                           GpuSchedulerT<eventql::Scheduler>, serving the
 *                                           data-node half of a distributed GROUP BY (every
 *                                           GroupByNode arrives marked partial, as
 *                                           EVQL_OP_QUERY_PARTIALAGGR delivers it): lowered
 *                                           plans run PartialGroupByExpression's GPU twin,
 *                                           the others eventql::Scheduler's own
 *                                           buildPartialGroupByExpression
 *   DUMP on|off                             include the compiled programs
 *   ROWS on|off                             include result rows (off: count only)
 *   SQL <statement>                         run it
 *   TIME <n> <statement>                    run it n times, report best seconds
 *   WRITE <file.cst> <nrows> <seed>         SURVEY 8c(ii) xorshift table through
 *                                           the reference's own CSTableWriter
 *   BUDGET <bytes>                          a fresh GPU table registry with this HBM budget
 *                                           (registered tables are forgotten)
 *   PARTITION <name> <dir> <file>:<skiplist>:<updates> ...
 *                                           a table whose scans are the reference's own
 *                                           eventql::PartitionCursor (server/sql/
 *                                           partition_cursor.cc) over a PartitionSnapshot
 *                                           with these lsm_tables (OLDEST first, as in
 *                                           PartitionState; <dir>/<file>.cst; the two flags
 *                                           are LSMTableRef::has_skiplist / has_updates);
 *                                           the same chain is registered with the GPU
 *                                           registry
 */
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <eventql/io/cstable/cstable_writer.h>
#include <eventql/sql/CSTableScan.h>
#include <eventql/sql/CSTableScanProvider.h>
#include <eventql/sql/qtree/GroupByNode.h>
#include <eventql/sql/qtree/SequentialScanNode.h>
#include <eventql/sql/result_cursor.h>
#include <eventql/sql/runtime/defaultruntime.h>
#include <eventql/sql/runtime/query_cache.h>
#include <eventql/sql/runtime/runtime.h>
#include <eventql/sql/runtime/tablerepository.h>
#include <eventql/sql/statements/select/groupby.h>
#include <eventql/util/SHA1.h>
#include <eventql/db/file_tracker.h>
#include <eventql/db/partition_snapshot.h>
#include <eventql/db/table.h>
#include <eventql/server/sql/partition_cursor.h>
#include <eventql/server/sql/scheduler.h>
#include <eventql/config/process_config.h>
#include "gpu_bridge.h"
#include "gpu_group_by_scan.h"
#include "gpu_partition.h"

using namespace evql_adapter;

namespace {

/* CSTableScanProvider whose scans are the Dremel-assembling csql::CSTableScan
 * (what PartitionCursor builds for nested / aggregating scans,
 * server/sql/partition_cursor.cc:206-213) */
struct DremelScanProvider : public csql::CSTableScanProvider {
  DremelScanProvider(const String& name, const String& file)
      : csql::CSTableScanProvider(name, file) {}
  Option<ScopedPtr<csql::TableExpression>> buildSequentialScan(
      csql::Transaction* txn, csql::ExecutionContext* ctx,
      RefPtr<csql::SequentialScanNode> node) const override {
    if (node->tableName() != table_name_) return None<ScopedPtr<csql::TableExpression>>();
    return Option<ScopedPtr<csql::TableExpression>>(ScopedPtr<csql::TableExpression>(
        new csql::CSTableScan(txn, ctx, node, cstable_file_)));
  }
};

/* A partition of an evqld table: the scan is the reference's own PartitionCursor over a
 * PartitionSnapshot (db/partition_snapshot.h:37-69), built exactly as
 * TableScan::openLocalPartition does (server/sql/table_scan.cc:116-144).  The schema
 * comes from the newest file (CSTableScanProvider::describe). */
struct PartitionProvider : public csql::CSTableScanProvider {
  PartitionProvider(const String& name, const String& newest_file,
                    RefPtr<eventql::PartitionSnapshot> snap)
      : csql::CSTableScanProvider(name, newest_file), snap_(snap) {}
  Option<ScopedPtr<csql::TableExpression>> buildSequentialScan(
      csql::Transaction* txn, csql::ExecutionContext* ctx,
      RefPtr<csql::SequentialScanNode> node) const override {
    if (node->tableName() != table_name_) return None<ScopedPtr<csql::TableExpression>>();
    return Option<ScopedPtr<csql::TableExpression>>(ScopedPtr<csql::TableExpression>(
        new eventql::PartitionCursor(txn, ctx, RefPtr<eventql::Table>(), snap_, node)));
  }
  RefPtr<eventql::PartitionSnapshot> snap_;
};

std::string jsonString(const char* s, size_t n) {
  std::string o = "\"";
  char buf[8];
  for (size_t i = 0; i < n; ++i) {
    unsigned char c = (unsigned char) s[i];
    if (c == '"' || c == '\\') {
      o += '\\';
      o += (char) c;
    } else if (c < 0x20 || c >= 0x7f) {
      snprintf(buf, sizeof(buf), "\\u%04x", c); /* bytes as latin-1 code points */
      o += buf;
    } else {
      o += (char) c;
    }
  }
  return o + "\"";
}
std::string jsonString(const std::string& s) { return jsonString(s.data(), s.size()); }

std::string hex(const void* p, size_t n) {
  static const char* d = "0123456789abcdef";
  std::string o;
  for (size_t i = 0; i < n; ++i) {
    unsigned char c = ((const unsigned char*) p)[i];
    o += d[c >> 4];
    o += d[c & 15];
  }
  return o;
}

/* one packed SVector element (svalue.cc:410-517) -> JSON; advances *cur */
std::string jsonCell(csql::SType t, const char** cur) {
  const char* p = *cur;
  char buf[64];
  std::string o;
  switch (t) {
    case csql::SType::NIL:
      *cur = p + 1;
      return "null";
    case csql::SType::UINT64:
    case csql::SType::TIMESTAMP64: {
      uint64_t v;
      memcpy(&v, p, 8);
      *cur = p + 9;
      if (p[8] & csql::STAG_NULL) return "null";
      snprintf(buf, sizeof(buf), "%llu", (unsigned long long) v);
      return buf;
    }
    case csql::SType::INT64: {
      int64_t v;
      memcpy(&v, p, 8);
      *cur = p + 9;
      if (p[8] & csql::STAG_NULL) return "null";
      snprintf(buf, sizeof(buf), "%lld", (long long) v);
      return buf;
    }
    case csql::SType::FLOAT64: {
      double v;
      uint64_t bits;
      memcpy(&v, p, 8);
      memcpy(&bits, p, 8);
      *cur = p + 9;
      if (p[8] & csql::STAG_NULL) return "null";
      /* exact: the IEEE bits as a hex string */
      snprintf(buf, sizeof(buf), "\"f:%016llx\"", (unsigned long long) bits);
      return buf;
    }
    case csql::SType::BOOL:
      *cur = p + 2;
      if (p[1] & csql::STAG_NULL) return "null";
      return p[0] ? "true" : "false";
    case csql::SType::STRING: {
      uint32_t len;
      memcpy(&len, p, 4);
      *cur = p + 4 + len + 1;
      if (p[4 + len] & csql::STAG_NULL) return "null";
      return jsonString(p + 4, len);
    }
  }
  return "null";
}

std::string jsonProgram(const csql::vm::Program* p) {
  LoweredProgram lp;
  std::string why;
  std::ostringstream o;
  if (!lowerProgram(p, &lp, &why)) {
    o << "{\"lowerable\":false,\"why\":" << jsonString(why) << ",\"return_type\":"
      << (int) p->return_type << ",\"n_instructions\":" << p->instructions.size() << "}";
    return o.str();
  }
  o << "{\"lowerable\":true,\"code\":[";
  for (size_t i = 0; i < lp.code.size(); ++i) {
    if (i) o << ",";
    o << "[" << lp.code[i].op << "," << lp.code[i].argt << "," << lp.code[i].arg0 << ","
      << jsonString(lp.symbols[i]) << "]";
  }
  o << "],\"method_call\":" << lp.c.method_call << ",\"method_accumulate\":"
    << lp.c.method_accumulate << ",\"return_type\":" << lp.c.return_type
    << ",\"aggregate_fn\":" << lp.c.aggregate_fn << ",\"static\":\""
    << hex(lp.literals.data(), lp.literals.size()) << "\"}";
  return o.str();
}

/* checks, for every call instruction, that the symbol table resolves the symbol
 * the adapter's table names to the very function pointer in the program */
bool symbolsAgree(csql::Transaction* txn, const csql::vm::Program* p, std::string* bad) {
  LoweredProgram lp;
  std::string why;
  if (!lowerProgram(p, &lp, &why)) return true;
  auto symtab = txn->getSymbolTable();
  for (size_t i = 0; i < lp.code.size(); ++i) {
    if (lp.code[i].op != EVQL_X_CALL_PURE) continue;
    if (lp.symbols[i] == "to_nil#nil/bool;") continue; /* two functions, one symbol */
    auto e = symtab->lookup(lp.symbols[i]);
    if (!e || (intptr_t) e->getFunction()->vtable.call != p->instructions[i].arg0) {
      *bad = lp.symbols[i];
      return false;
    }
  }
  return true;
}

struct ProbeState {
  bool dump = false;
  bool rows = true;
  std::string mode = "cpu";
  bool partial = false;
  bool strict = false;
  std::string programs_json; /* filled by the scheduler while the plan is built */
  std::string decisions_json;
};
ProbeState g_state;

/* what the probe reads back from whichever scheduler is installed */
struct DecisionSource {
  virtual ~DecisionSource() {}
  virtual const std::vector<GpuLowering::Decision>& probeDecisions() const = 0;
  virtual void probeClear() = 0;
  virtual uint64_t probeCacheHits() const = 0;
};

/* GpuScheduler with (a) a program dump of what the reference compiles for the
 * GROUP BY + scan and (b) the CPU PartialGroupByExpression of a data node
 * (server/sql/scheduler.cc:79-115) when asked for */
class ProbeScheduler : public GpuScheduler, public DecisionSource {
public:
  ProbeScheduler(std::shared_ptr<GpuTableRegistry> t, GpuSchedulerOptions o, bool cpu_partial)
      : GpuScheduler(t, o), cpu_partial_(cpu_partial) {}
  const std::vector<GpuLowering::Decision>& probeDecisions() const override { return decisions(); }
  void probeClear() override { clearDecisions(); }
  uint64_t probeCacheHits() const override { return queryCacheHits(); }

protected:
  ScopedPtr<csql::TableExpression> buildGroupByExpression(
      csql::Transaction* txn, csql::ExecutionContext* ctx,
      RefPtr<csql::GroupByNode> node) override {
    if (g_state.dump) dumpPrograms(txn, node.get());
    if (cpu_partial_) {
      Vector<csql::ValueExpression> select_expressions, group_expressions;
      SHA1Hash fingerprint;
      for (const auto& sl : node->selectList()) {
        select_expressions.emplace_back(
            txn->getCompiler()->buildValueExpression(txn, sl->expression()));
        fingerprint = SHA1::compute(fingerprint.toString() + sl->toString());
      }
      for (const auto& e : node->groupExpressions()) {
        group_expressions.emplace_back(txn->getCompiler()->buildValueExpression(txn, e));
        fingerprint = SHA1::compute(fingerprint.toString() + e->toString());
      }
      return mkScoped(new csql::PartialGroupByExpression(
          txn, std::move(select_expressions), std::move(group_expressions), fingerprint,
          buildTableExpression(txn, ctx,
                               node->inputTable().asInstanceOf<csql::TableExpressionNode>())));
    }
    return GpuScheduler::buildGroupByExpression(txn, ctx, node);
  }

  void dumpPrograms(csql::Transaction* txn, csql::GroupByNode* node) {
    std::ostringstream o;
    std::string bad;
    bool agree = true;
    auto emit = [&](RefPtr<csql::ValueExpressionNode> e) {
      csql::ValueExpression c = txn->getCompiler()->buildValueExpression(txn, e);
      agree = agree && symbolsAgree(txn, c.program(), &bad);
      return jsonProgram(c.program());
    };
    o << "{\"select\":[";
    bool first = true;
    for (const auto& sl : node->selectList()) {
      o << (first ? "" : ",") << emit(sl->expression());
      first = false;
    }
    o << "],\"group\":[";
    first = true;
    for (const auto& e : node->groupExpressions()) {
      o << (first ? "" : ",") << emit(e);
      first = false;
    }
    o << "]";
    auto seqscan = dynamic_cast<csql::SequentialScanNode*>(node->inputTable().get());
    if (seqscan) {
      o << ",\"scan_columns\":[";
      auto cols = seqscan->selectedColumns();
      for (size_t i = 0; i < cols.size(); ++i) {
        o << (i ? "," : "") << "[" << jsonString(cols[i]) << ","
          << (int) seqscan->getInputColumnType(i) << "]";
      }
      o << "],\"scan_select\":[";
      first = true;
      for (const auto& sl : seqscan->selectList()) {
        o << (first ? "" : ",") << emit(sl->expression());
        first = false;
      }
      o << "],\"where\":";
      auto w = seqscan->whereExpression();
      if (w.isEmpty()) {
        o << "null";
      } else {
        o << emit(w.get());
      }
      o << ",\"aggregation_strategy\":" << (int) seqscan->aggregationStrategy();
    }
    o << ",\"symbols_agree\":" << (agree ? "true" : "false");
    if (!agree) o << ",\"bad_symbol\":" << jsonString(bad);
    o << "}";
    g_state.programs_json = o.str();
  }

  bool cpu_partial_;
};

/* The object evqld would hand to Runtime::setScheduler: the GPU mix-in over
 * eventql::Scheduler (server/sql/scheduler.h:37).  The probe plays the data node of a
 * distributed GROUP BY: the coordinator's eventql::Scheduler::buildPipelineGroupByExpression
 * ships GroupByNode copies with setIsPartialAggreagtion(true) (server/sql/scheduler.cc:
 * 146-158) -- the coordinator half itself needs a running cluster (PartitionMap, metadata
 * client, RPC) and is not reachable from a standalone csql::Runtime. */
class ProbeEvqldScheduler : public GpuSchedulerT<eventql::Scheduler>, public DecisionSource {
public:
  ProbeEvqldScheduler(std::shared_ptr<GpuTableRegistry> t, GpuSchedulerOptions o,
                      eventql::ProcessConfig* config)
      : GpuSchedulerT<eventql::Scheduler>(t, o, config, (eventql::PartitionMap*) nullptr,
                                          (eventql::ConfigDirectory*) nullptr,
                                          (eventql::InternalAuth*) nullptr) {}
  const std::vector<GpuLowering::Decision>& probeDecisions() const override { return decisions(); }
  void probeClear() override { clearDecisions(); }
  uint64_t probeCacheHits() const override { return queryCacheHits(); }

protected:
  ScopedPtr<csql::TableExpression> buildGroupByExpression(
      csql::Transaction* txn, csql::ExecutionContext* ctx,
      RefPtr<csql::GroupByNode> node) override {
    node->setIsPartialAggreagtion(true);
    return GpuSchedulerT<eventql::Scheduler>::buildGroupByExpression(txn, ctx, node);
  }
};

struct Probe {
  RefPtr<csql::Runtime> runtime;
  std::shared_ptr<GpuTableRegistry> registry;
  struct Tbl {
    std::string name, file, kind;
    RefPtr<eventql::PartitionSnapshot> snap; /* kind == "partition" */
  };
  /* what a PartitionSnapshot's constructor touches of the server (file refcounts) */
  std::unique_ptr<eventql::FileTracker> file_tracker;
  eventql::DatabaseContext dbctx{};
  std::vector<Tbl> tables;
  DecisionSource* scheduler = nullptr; /* owned by the runtime */
  RefPtr<eventql::ProcessConfig> evqld_config;

  Probe() {
    runtime = csql::Runtime::getDefaultRuntime();
    registry = std::make_shared<GpuTableRegistry>(0);
  }

  void installScheduler() {
    GpuSchedulerOptions o;
    o.lower_group_by = g_state.mode == "gpu";
    o.lower_scans = g_state.mode == "gpuscan";
    o.partial = g_state.partial;
    o.strict = g_state.strict;
    bool cpu_partial = g_state.partial && g_state.mode == "cpu";
    if (g_state.mode == "evqld") {
      if (!evqld_config.get()) {
        eventql::ProcessConfigBuilder b;
        b.setProperty("server.query_max_concurrent_shards", "8");
        b.setProperty("server.query_max_concurrent_shards_per_host", "6");
        b.setProperty("server.query_failed_shard_policy", "tolerate");
        evqld_config = b.getConfig();
      }
      o.lower_group_by = true;
      auto s = new ProbeEvqldScheduler(registry, o, evqld_config.get());
      scheduler = s;
      runtime->setScheduler(ScopedPtr<csql::Scheduler>(s));
      return;
    }
    auto s = new ProbeScheduler(registry, o, cpu_partial);
    scheduler = s;
    runtime->setScheduler(ScopedPtr<csql::Scheduler>(s));
  }

  std::string runOnce(const std::string& sql, bool want_rows, double* seconds) {
    std::ostringstream o;
    g_state.programs_json.clear();
    if (scheduler) scheduler->probeClear();
    auto t0 = std::chrono::steady_clock::now();
    o << "{\"sql\":" << jsonString(sql) << ",\"mode\":" << jsonString(g_state.mode)
      << ",\"partial\":" << (g_state.partial ? "true" : "false");
    try {
      auto txn = runtime->newTransaction();
      auto repo = mkScoped(new csql::TableRepository());
      for (const auto& t : tables) {
        if (t.kind == "partition") {
          repo->addProvider(new PartitionProvider(t.name, t.file, t.snap));
        } else if (t.kind == "dremel") {
          repo->addProvider(new DremelScanProvider(t.name, t.file));
        } else {
          repo->addProvider(new csql::CSTableScanProvider(t.name, t.file));
        }
      }
      txn->setTableProvider(repo.release());
      auto qplan = runtime->buildQueryPlan(txn.get(), sql);
      auto cursor = qplan->execute(0);
      size_t ncols = cursor->getColumnCount();
      std::vector<csql::SType> types;
      o << ",\"ok\":true,\"types\":[";
      for (size_t i = 0; i < ncols; ++i) {
        types.push_back(cursor->getColumnType(i));
        o << (i ? "," : "") << (int) types[i];
      }
      o << "],\"columns\":[";
      const auto& names = qplan->getStatementgetResultColumns(0);
      for (size_t i = 0; i < names.size(); ++i) o << (i ? "," : "") << jsonString(names[i]);
      o << "]";
      size_t nrows = 0;
      std::ostringstream rows;
      /* batch-wise, straight from the SVector buffers (result_cursor.cc:114-133) */
      while (cursor->isValid()) {
        size_t n = cursor->getBufferCount();
        std::vector<const char*> cur(ncols);
        for (size_t c = 0; c < ncols; ++c) cur[c] = (const char*) cursor->getColumnBuffer(c);
        for (size_t r = 0; r < n; ++r) {
          if (want_rows) rows << (nrows + r ? "," : "") << "[";
          for (size_t c = 0; c < ncols; ++c) {
            std::string cell = jsonCell(types[c], &cur[c]);
            if (want_rows) rows << (c ? "," : "") << cell;
          }
          if (want_rows) rows << "]";
        }
        nrows += n;
        auto rc = cursor->nextBatch();
        if (!rc.isSuccess()) RAISE(kRuntimeError, rc.getMessage());
      }
      o << ",\"nrows\":" << nrows;
      if (want_rows) o << ",\"rows\":[" << rows.str() << "]";
    } catch (const std::exception& e) {
      o << ",\"ok\":false,\"error\":" << jsonString(e.what());
    }
    auto t1 = std::chrono::steady_clock::now();
    *seconds = std::chrono::duration<double>(t1 - t0).count();
    if (g_state.dump && !g_state.programs_json.empty()) {
      o << ",\"programs\":" << g_state.programs_json;
    }
    if (scheduler && !scheduler->probeDecisions().empty()) {
      o << ",\"decisions\":[";
      bool first = true;
      for (const auto& d : scheduler->probeDecisions()) {
        o << (first ? "" : ",") << "{\"node\":" << jsonString(d.node) << ",\"lowered\":"
          << (d.lowered ? "true" : "false") << ",\"reason\":" << jsonString(d.reason) << "}";
        first = false;
      }
      o << "]";
      o << ",\"query_cache_hits\":" << scheduler->probeCacheHits();
      o << ",\"resident_bytes\":" << registry->residentBytes();
    }
    return o.str();
  }
};

/* the xorshift64 table, written through the reference's own
 * CSTableWriter (io/cstable/cstable_writer.cc:46-293) */
void writeSurveyTable(const std::string& file, uint64_t nrows, uint64_t seed) {
  using namespace cstable;
  TableSchema schema;
  schema.addUnsignedInteger("k", false, ColumnEncoding::UINT64_PLAIN);
  schema.addUnsignedInteger("a", false, ColumnEncoding::UINT64_PLAIN);
  schema.addUnsignedInteger("b", false, ColumnEncoding::UINT64_PLAIN);
  schema.addFloat("v", false);
  auto w = CSTableWriter::createFile(file, schema);
  auto ck = w->getColumnWriter("k"), ca = w->getColumnWriter("a"), cb = w->getColumnWriter("b"),
       cv = w->getColumnWriter("v");
  uint64_t x = seed;
  for (uint64_t i = 0; i < nrows; ++i) {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    ck->writeUnsignedInt(0, 0, x % 1000);
    ca->writeUnsignedInt(0, 0, (x >> 8) & 0xffff);
    cb->writeUnsignedInt(0, 0, (x >> 24) & 0xffff);
    cv->writeFloat(0, 0, (double) (x >> 40) / 1024.0);
    w->addRow();
  }
  w->commit();
}

}  // namespace

int main(int argc, char** argv) {
  Probe probe;
  probe.installScheduler();
  std::string line;
  while (std::getline(std::cin, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream is(line);
    std::string cmd;
    is >> cmd;
    if (cmd == "TABLE") {
      Probe::Tbl t;
      is >> t.name >> t.file;
      if (!(is >> t.kind)) t.kind = "fast";
      std::string tag;
      is >> tag;
      bool replaced = false;
      for (auto& e : probe.tables) {
        if (e.name == t.name) {
          e = t;
          replaced = true;
        }
      }
      if (!replaced) probe.tables.push_back(t);
      probe.registry->registerTable(t.name, t.file,
                                    t.kind == "dremel" ? ScanKind::DREMEL : ScanKind::FAST, tag);
    } else if (cmd == "PARTITION") {
      Probe::Tbl t;
      std::string dir, spec;
      is >> t.name >> dir;
      t.kind = "partition";
      if (!probe.file_tracker) {
        probe.file_tracker.reset(new eventql::FileTracker(dir));
        probe.dbctx.file_tracker = probe.file_tracker.get();
      }
      eventql::PartitionState state;
      state.set_tsdb_namespace("probe");
      state.set_partition_key(SHA1::compute(t.name).data(), SHA1Hash::kSize);
      state.set_table_key(t.name);
      uint64_t seq = 1;
      while (is >> spec) {
        /* <file>:<has_skiplist>:<has_updates>, oldest first like PartitionState::lsm_tables */
        size_t c2 = spec.rfind(':'), c1 = spec.rfind(':', c2 - 1);
        std::string fname = spec.substr(0, c1);
        bool skiplist = spec[c1 + 1] == '1', updates = spec[c2 + 1] == '1';
        auto ref = state.add_lsm_tables();
        ref->set_filename(fname);
        ref->set_first_sequence(seq);
        ref->set_last_sequence(seq);
        ref->set_has_skiplist(skiplist);
        ref->set_has_updates(updates);
        ++seq;
        t.file = dir + "/" + fname + ".cst"; /* ends as the newest */
      }
      state.set_lsm_sequence(seq);
      t.snap = mkRef(new eventql::PartitionSnapshot(state, dir, "", &probe.dbctx, 0));
      bool replaced = false;
      for (auto& e : probe.tables) {
        if (e.name == t.name) {
          e = t;
          replaced = true;
        }
      }
      if (!replaced) probe.tables.push_back(t);
      /* the GPU registry finds the chain the way evqld's would: through a resolver that
       * maps the scan's table name to the partition's snapshot (gpu_partition.h) */
      Probe* pp = &probe;
      probe.registry->setResolver(partitionResolver(
          [pp](const std::string& name) -> RefPtr<eventql::PartitionSnapshot> {
            for (const auto& e : pp->tables) {
              if (e.name == name && e.kind == "partition") return e.snap;
            }
            return RefPtr<eventql::PartitionSnapshot>();
          }));
    } else if (cmd == "BUDGET") {
      unsigned long long bytes = 0;
      is >> bytes;
      probe.registry = std::make_shared<GpuTableRegistry>(0, (uint64_t) bytes);
      probe.tables.clear();
      probe.installScheduler();
    } else if (cmd == "CACHE") {
      std::string dir;
      is >> dir;
      /* (lives as long as the process, like evqld's) */
      probe.runtime->setQueryCache(new csql::QueryCache(dir, 8192, 0));
    } else if (cmd == "MODE") {
      is >> g_state.mode;
      g_state.partial = g_state.strict = false;
      std::string f;
      while (is >> f) {
        if (f == "partial") g_state.partial = true;
        if (f == "strict") g_state.strict = true;
      }
      probe.installScheduler();
    } else if (cmd == "DUMP") {
      std::string f;
      is >> f;
      g_state.dump = f == "on";
    } else if (cmd == "ROWS") {
      std::string f;
      is >> f;
      g_state.rows = f == "on";
    } else if (cmd == "SQL") {
      std::string sql;
      std::getline(is, sql);
      double s;
      std::string out = probe.runOnce(sql.substr(sql.find_first_not_of(' ')), g_state.rows, &s);
      printf("%s,\"seconds\":%.6f}\n", out.c_str(), s);
      fflush(stdout);
    } else if (cmd == "TIME") {
      int n;
      is >> n;
      std::string sql;
      std::getline(is, sql);
      sql = sql.substr(sql.find_first_not_of(' '));
      double best = 1e300, s;
      std::string out;
      for (int i = 0; i < n; ++i) {
        out = probe.runOnce(sql, false, &s);
        if (s < best) best = s;
      }
      printf("%s,\"seconds\":%.6f,\"runs\":%d}\n", out.c_str(), best, n);
      fflush(stdout);
    } else if (cmd == "WRITE") {
      std::string file;
      uint64_t nrows, seed;
      is >> file >> nrows >> seed;
      try {
        writeSurveyTable(file, nrows, seed);
        printf("{\"written\":%s,\"nrows\":%llu}\n", jsonString(file).c_str(),
               (unsigned long long) nrows);
      } catch (const std::exception& e) {
        printf("{\"ok\":false,\"error\":%s}\n", jsonString(e.what()).c_str());
      }
      fflush(stdout);
    } else {
      fprintf(stderr, "unknown command: %s\n", cmd.c_str());
      return 2;
    }
  }
  return 0;
}
