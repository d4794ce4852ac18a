/*
 * gpu_bridge.h -- csql::vm::Program -> evql_program_t.
 *
 * REFERENCE-SIDE ADAPTER: this translation unit includes the reference's headers
 * (sql/runtime/vm.h, sql/expressions/*.h) and therefore compiles only inside a
 * reference build (in this repository: the test-only reference build, where /root/reference exists).
 * It is the code a maintainer adds under src/eventql/sql/runtime/ to bind
 * libevql_mi355x.so (include/evql_gpu.h); nothing in the product library depends
 * on it.
 *
 * evql_instr_t mirrors vm::Instruction (sql/runtime/vm.h:54-60) field for field;
 * only arg0 changes meaning where the reference stores a pointer:
 *   X_CALL_PURE      fn pointer            -> EVQL_FN(family, type slot)
 *   X_CALL_INSTANCE  accumulate/get pointer-> EVQL_INSTANCE_ACCUMULATE / _GET
 *   X_LITERAL        pointer into Program::static_storage -> byte offset into a
 *                    copied literal pool (value bytes + tag, svalue.cc:533-549)
 */
#pragma once
#include <string>
#include <vector>
#include "eventql/eventql.h"
#include <eventql/util/io/inputstream.h>
#include <eventql/util/io/outputstream.h>
#include <eventql/sql/svalue.h>
#include <eventql/sql/runtime/vm.h> /* not self-contained: needs the four above */
#include "evql_gpu.h"

namespace evql_adapter {

struct LoweredProgram {
  std::vector<evql_instr_t> code;
  std::vector<uint8_t> literals;
  /* symbol ("name#ret/arg;arg;", runtime/symboltable.cc:33-41) of every call
   * instruction, "" for the others; for tests and diagnostics */
  std::vector<std::string> symbols;
  evql_program_t c;

  LoweredProgram() { c = evql_program_t(); }
  LoweredProgram(const LoweredProgram&) = delete;
  LoweredProgram& operator=(const LoweredProgram&) = delete;
  void seal();  /* (re)points c at the vectors */
};

/* false => the program calls a function outside the lowerable set (SURVEY 8a
 * "Lowerable op table"); *why receives its address-less description */
bool lowerProgram(const csql::vm::Program* p, LoweredProgram* out, std::string* why);

/* id -> reference symbol, e.g. EVQL_FN(EVQL_FAM_GT, EVQL_TS_UINT64) ->
 * "gt#bool/uint64;uint64;" */
std::string pureFunctionSymbol(int64_t fn_id);
std::string aggregateSymbol(uint32_t aggregate_fn);

}  // namespace evql_adapter
