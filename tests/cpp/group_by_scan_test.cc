// group_by_scan_test.cc -- drives the operator exactly the way the reference's
// engine drives a csql::TableExpression (ResultCursor: execute() once, then
// nextBatch() until *len == 0) and prints the result in the format of the
// reference's SQL golden files (test/sql/*.result.txt: header line, then
// ';'-separated cells).
//
//   group_by_scan_test <file.cst>
//
// Query: select k, sum(a), count(1) from t where a > 30000 group by k
// hand-assembled as the vm::Program bytecode Compiler::compile would emit
// (sql/runtime/compiler.cc:50-248).
#include <cinttypes>
#include <cstdio>
#include <map>
#include "../../include/evql_host.hpp"

using namespace evql_host;

static evql_program_t program(const std::vector<evql_instr_t>& code, uint32_t acc, uint32_t rtype,
                              uint32_t aggfn, const std::vector<uint8_t>& lits) {
  evql_program_t p;
  p.code = code.data();
  p.code_len = uint32_t(code.size());
  p.method_call = 0;
  p.method_accumulate = acc;
  p.return_type = rtype;
  p.aggregate_fn = aggfn;
  p.static_storage = lits.data();
  p.static_storage_len = lits.size();
  return p;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s file.cst\n", argv[0]);
    return 2;
  }
  evql_ctx_t* ctx = nullptr;
  if (evql_ctx_create(0, nullptr, &ctx) != EVQL_OK) {
    fprintf(stderr, "ctx: %s\n", evql_last_error());
    return 1;
  }
  evql_table_t* table = nullptr;
  if (evql_table_open_file(ctx, argv[1], &table) != EVQL_OK) {
    fprintf(stderr, "open: %s\n", evql_last_error());
    return 1;
  }

  // scan columns: a (WHERE first), k
  const char* scan_cols[] = {"a", "k"};
  const uint32_t scan_types[] = {EVQL_T_UINT64, EVQL_T_UINT64};
  // WHERE a > 30000
  std::vector<uint8_t> lit30000(9, 0);
  uint64_t v = 30000;
  memcpy(lit30000.data(), &v, 8);
  std::vector<evql_instr_t> where_code = {
      {EVQL_X_INPUT, EVQL_T_UINT64, 0},
      {EVQL_X_LITERAL, EVQL_T_UINT64, 0},
      {EVQL_X_CALL_PURE, 0, EVQL_FN(EVQL_FAM_GT, EVQL_TS_UINT64)},
      {EVQL_X_RETURN, 0, 0}};
  evql_program_t where = program(where_code, 0, EVQL_T_BOOL, EVQL_AGG_NONE, lit30000);
  // scan select list (bare refs, first use order at the GROUP BY level): k, a
  std::vector<uint8_t> nolit(1, 0);
  std::vector<evql_instr_t> ref_k = {{EVQL_X_INPUT, EVQL_T_UINT64, 1}, {EVQL_X_RETURN, 0, 0}};
  std::vector<evql_instr_t> ref_a = {{EVQL_X_INPUT, EVQL_T_UINT64, 0}, {EVQL_X_RETURN, 0, 0}};
  evql_program_t scan_select[] = {program(ref_k, 0, EVQL_T_UINT64, EVQL_AGG_NONE, nolit),
                                  program(ref_a, 0, EVQL_T_UINT64, EVQL_AGG_NONE, nolit)};
  // group by k  (X_INPUT 0 of the scan output)
  std::vector<evql_instr_t> g_k = {{EVQL_X_INPUT, EVQL_T_UINT64, 0}, {EVQL_X_RETURN, 0, 0}};
  evql_program_t group[] = {program(g_k, 0, EVQL_T_UINT64, EVQL_AGG_NONE, nolit)};
  // select k, sum(a), count(1)
  std::vector<evql_instr_t> s_sum = {{EVQL_X_CALL_INSTANCE, 0, EVQL_INSTANCE_GET},
                                     {EVQL_X_RETURN, 0, 0},
                                     {EVQL_X_INPUT, EVQL_T_UINT64, 1},
                                     {EVQL_X_CALL_INSTANCE, 0, EVQL_INSTANCE_ACCUMULATE},
                                     {EVQL_X_RETURN, 0, 0}};
  std::vector<uint8_t> lit1(9, 0);
  lit1[0] = 1;
  std::vector<evql_instr_t> s_cnt = {{EVQL_X_CALL_INSTANCE, 0, EVQL_INSTANCE_GET},
                                     {EVQL_X_RETURN, 0, 0},
                                     {EVQL_X_LITERAL, EVQL_T_UINT64, 0},
                                     {EVQL_X_CALL_PURE, 0, EVQL_FN(EVQL_FAM_TO_NIL, EVQL_TS_UINT64)},
                                     {EVQL_X_CALL_INSTANCE, 0, EVQL_INSTANCE_ACCUMULATE},
                                     {EVQL_X_RETURN, 0, 0}};
  evql_program_t select[] = {program(g_k, 0, EVQL_T_UINT64, EVQL_AGG_NONE, nolit),
                             program(s_sum, 2, EVQL_T_UINT64, EVQL_AGG_SUM_UINT64, nolit),
                             program(s_cnt, 2, EVQL_T_UINT64, EVQL_AGG_COUNT, lit1)};

  evql_plan_desc_t plan;
  memset(&plan, 0, sizeof(plan));
  plan.scan_columns = scan_cols;
  plan.scan_column_types = scan_types;
  plan.n_scan_columns = 2;
  plan.where = &where;
  plan.scan_select = scan_select;
  plan.n_scan_select = 2;
  plan.group_exprs = group;
  plan.n_group = 1;
  plan.select_exprs = select;
  plan.n_select = 3;
  plan.group_mode = EVQL_MODE_FINAL;
  plan.scan_mode = EVQL_SCAN_FLAT;

  int heartbeats = 0;
  if (argc >= 3 && std::string(argv[2]) == "merge") {
    // PartialGroupBy over two row ranges on the device -> GroupByMerge
    // (scheduler.cc:117-162 fan-out, groupby.cc:528-672 merge)
    try {
      GroupByMerge merge(plan);
      const uint64_t n = evql_table_num_rows(table), m = n / 3;
      for (int part = 0; part < 2; ++part) {
        evql_plan_desc_t pp = plan;
        pp.group_mode = EVQL_MODE_PARTIAL;
        pp.row_begin = part == 0 ? 0 : m;
        pp.row_end = part == 0 ? m : n;
        GpuGroupByScan partial(ctx, table, pp);
        if (partial.getColumnCount() != 2 || partial.getColumnType(0) != SType::STRING) {
          fprintf(stderr, "bad partial column metadata\n");
          return 1;
        }
        ReturnCode rc = merge.addPart(&partial);
        if (!rc.isSuccess()) {
          fprintf(stderr, "addPart: %s\n", rc.getMessage().c_str());
          return 1;
        }
      }
      ResultCursor cursor(&merge);
      std::map<uint64_t, std::pair<uint64_t, uint64_t>> rows;
      while (cursor.nextBatch()) {
        for (size_t i = 0; i < cursor.batchLength(); ++i) {
          uint64_t k, s, cnt;
          memcpy(&k, static_cast<const char*>(cursor.column(0).getData()) + 9 * i, 8);
          memcpy(&s, static_cast<const char*>(cursor.column(1).getData()) + 9 * i, 8);
          memcpy(&cnt, static_cast<const char*>(cursor.column(2).getData()) + 9 * i, 8);
          rows[k] = {s, cnt};
        }
      }
      printf("k;sum(a);count(1)\n");
      for (const auto& r : rows) {
        printf("%" PRIu64 ";%" PRIu64 ";%" PRIu64 "\n", r.first, r.second.first, r.second.second);
      }
    } catch (const std::exception& e) {
      fprintf(stderr, "error: %s\n", e.what());
      return 1;
    }
    evql_table_close(table);
    evql_ctx_destroy(ctx);
    return 0;
  }
  try {
    GpuGroupByScan op(ctx, table, plan, [&heartbeats]() {
      ++heartbeats;
      return ReturnCode::success();
    });
    if (op.getColumnCount() != 3 || op.getColumnType(1) != SType::UINT64) {
      fprintf(stderr, "bad column metadata\n");
      return 1;
    }
    ResultCursor cursor(&op);
    std::map<uint64_t, std::pair<uint64_t, uint64_t>> rows;  // ordered output
    size_t batches = 0;
    while (cursor.nextBatch()) {
      ++batches;
      const size_t n = cursor.batchLength();
      if (n > kOutputBatchSize) {
        fprintf(stderr, "batch too large\n");
        return 1;
      }
      for (size_t c = 0; c < 3; ++c) {
        if (cursor.column(c).getSize() != n * 9) {
          fprintf(stderr, "unexpected packed size\n");
          return 1;
        }
      }
      for (size_t i = 0; i < n; ++i) {
        uint64_t k, s, cnt;
        memcpy(&k, static_cast<const char*>(cursor.column(0).getData()) + 9 * i, 8);
        memcpy(&s, static_cast<const char*>(cursor.column(1).getData()) + 9 * i, 8);
        memcpy(&cnt, static_cast<const char*>(cursor.column(2).getData()) + 9 * i, 8);
        rows[k] = {s, cnt};
      }
    }
    if (cursor.nextBatch()) {
      fprintf(stderr, "EOF must be sticky\n");
      return 1;
    }
    printf("k;sum(a);count(1)\n");
    for (const auto& r : rows) {
      printf("%" PRIu64 ";%" PRIu64 ";%" PRIu64 "\n", r.first, r.second.first, r.second.second);
    }
    fprintf(stderr, "batches=%zu heartbeats=%d\n", batches, heartbeats);
    if (heartbeats < 1) return 1;
  } catch (const std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  evql_table_close(table);
  evql_ctx_destroy(ctx);
  return 0;
}
