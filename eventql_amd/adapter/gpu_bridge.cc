/*
 * gpu_bridge.cc -- see gpu_bridge.h.  Function identity: Compiler::compileMethodCall
 * (sql/runtime/compiler.cc:211-246) stores SFunction::vtable.call / .get in
 * Instruction::arg0 and Compiler::compile (:67-100) vtable.accumulate; the
 * SymbolTable holds copies of the SFunction constants declared in
 * sql/expressions/{boolean,math,conversion,aggregate}.h (registered in
 * sql/defaults.cc:37-169), so the pointers compare equal to those constants'.
 */
#include "gpu_bridge.h"
#include <map>
#include <string.h>
#include <eventql/sql/SFunction.h>
#include <eventql/sql/expressions/aggregate.h>
#include <eventql/sql/expressions/boolean.h>
#include <eventql/sql/expressions/conversion.h>
#include <eventql/sql/expressions/math.h>
#include <eventql/sql/expressions/string.h>

namespace evql_adapter {
namespace ex = csql::expressions;

namespace {

struct PureEntry {
  const csql::SFunction* fn;
  int64_t id;
  const char* symbol;
};

#define EVQL_CMP_ROWS(name, FAM)                                                          \
  {&ex::name##_uint64, EVQL_FN(FAM, EVQL_TS_UINT64), #name "#bool/uint64;uint64;"},        \
  {&ex::name##_int64, EVQL_FN(FAM, EVQL_TS_INT64), #name "#bool/int64;int64;"},            \
  {&ex::name##_float64, EVQL_FN(FAM, EVQL_TS_FLOAT64), #name "#bool/float64;float64;"},    \
  {&ex::name##_string, EVQL_FN(FAM, EVQL_TS_STRING), #name "#bool/string;string;"},        \
  {&ex::name##_timestamp64, EVQL_FN(FAM, EVQL_TS_TIMESTAMP64),                             \
   #name "#bool/timestamp64;timestamp64;"}

#define EVQL_ARITH_ROWS(name, FAM)                                                        \
  {&ex::name##_uint64, EVQL_FN(FAM, EVQL_TS_UINT64), #name "#uint64/uint64;uint64;"},      \
  {&ex::name##_int64, EVQL_FN(FAM, EVQL_TS_INT64), #name "#int64/int64;int64;"},           \
  {&ex::name##_float64, EVQL_FN(FAM, EVQL_TS_FLOAT64), #name "#float64/float64;float64;"}

const PureEntry kPure[] = {
    /* sql/expressions/boolean.h (boolean.cc:38-713) */
    {&ex::logical_and, EVQL_FN(EVQL_FAM_LOGICAL_AND, 0), "logical_and#bool/bool;bool;"},
    {&ex::logical_or, EVQL_FN(EVQL_FAM_LOGICAL_OR, 0), "logical_or#bool/bool;bool;"},
    {&ex::neg, EVQL_FN(EVQL_FAM_NEG, 0), "neg#bool/bool;"},
    {&ex::cmp_uint64, EVQL_FN(EVQL_FAM_CMP, EVQL_TS_UINT64), "cmp#int64/uint64;uint64;"},
    {&ex::cmp_int64, EVQL_FN(EVQL_FAM_CMP, EVQL_TS_INT64), "cmp#int64/int64;int64;"},
    {&ex::cmp_float64, EVQL_FN(EVQL_FAM_CMP, EVQL_TS_FLOAT64), "cmp#int64/float64;float64;"},
    {&ex::cmp_string, EVQL_FN(EVQL_FAM_CMP, EVQL_TS_STRING), "cmp#int64/string;string;"},
    {&ex::cmp_timestamp64, EVQL_FN(EVQL_FAM_CMP, EVQL_TS_TIMESTAMP64),
     "cmp#int64/timestamp64;timestamp64;"},
    EVQL_CMP_ROWS(eq, EVQL_FAM_EQ),
    {&ex::eq_bool, EVQL_FN(EVQL_FAM_EQ, EVQL_TS_BOOL), "eq#bool/bool;bool;"},
    EVQL_CMP_ROWS(neq, EVQL_FAM_NEQ),
    {&ex::neq_bool, EVQL_FN(EVQL_FAM_NEQ, EVQL_TS_BOOL), "neq#bool/bool;bool;"},
    EVQL_CMP_ROWS(lt, EVQL_FAM_LT),
    EVQL_CMP_ROWS(lte, EVQL_FAM_LTE),
    EVQL_CMP_ROWS(gt, EVQL_FAM_GT),
    EVQL_CMP_ROWS(gte, EVQL_FAM_GTE),
    /* sql/expressions/math.h (math.cc:34-251) */
    EVQL_ARITH_ROWS(add, EVQL_FAM_ADD),
    EVQL_ARITH_ROWS(sub, EVQL_FAM_SUB),
    EVQL_ARITH_ROWS(mul, EVQL_FAM_MUL),
    EVQL_ARITH_ROWS(div, EVQL_FAM_DIV),
    EVQL_ARITH_ROWS(mod, EVQL_FAM_MOD),
    EVQL_ARITH_ROWS(pow, EVQL_FAM_POW),
    /* sql/expressions/conversion.h (conversion.cc:34-245); the type slot is the
     * ARGUMENT's.  to_nil_string is declared with a BOOL argument in the reference
     * (conversion.cc:76-79), so its symbol equals to_nil_bool's; the function
     * pointer still tells them apart */
    {&ex::to_nil_uint64, EVQL_FN(EVQL_FAM_TO_NIL, EVQL_TS_UINT64), "to_nil#nil/uint64;"},
    {&ex::to_nil_int64, EVQL_FN(EVQL_FAM_TO_NIL, EVQL_TS_INT64), "to_nil#nil/int64;"},
    {&ex::to_nil_float64, EVQL_FN(EVQL_FAM_TO_NIL, EVQL_TS_FLOAT64), "to_nil#nil/float64;"},
    {&ex::to_nil_bool, EVQL_FN(EVQL_FAM_TO_NIL, EVQL_TS_BOOL), "to_nil#nil/bool;"},
    {&ex::to_nil_string, EVQL_FN(EVQL_FAM_TO_NIL, EVQL_TS_STRING), "to_nil#nil/bool;"},
    {&ex::to_nil_timestamp64, EVQL_FN(EVQL_FAM_TO_NIL, EVQL_TS_TIMESTAMP64),
     "to_nil#nil/timestamp64;"},
    {&ex::to_int64_uint64, EVQL_FN(EVQL_FAM_TO_INT64, EVQL_TS_UINT64), "to_int64#int64/uint64;"},
    {&ex::to_int64_float64, EVQL_FN(EVQL_FAM_TO_INT64, EVQL_TS_FLOAT64),
     "to_int64#int64/float64;"},
    {&ex::to_int64_bool, EVQL_FN(EVQL_FAM_TO_INT64, EVQL_TS_BOOL), "to_int64#int64/bool;"},
    {&ex::to_int64_timestamp64, EVQL_FN(EVQL_FAM_TO_INT64, EVQL_TS_TIMESTAMP64),
     "to_int64#int64/timestamp64;"},
    {&ex::to_timestamp64_int64, EVQL_FN(EVQL_FAM_TO_TIMESTAMP64, EVQL_TS_INT64),
     "to_timestamp64#timestamp64/int64;"},
    {&ex::to_timestamp64_float64, EVQL_FN(EVQL_FAM_TO_TIMESTAMP64, EVQL_TS_FLOAT64),
     "to_timestamp64#timestamp64/float64;"},
    /* strings: conversion.cc:140-215, expressions/string.cc (select-list only; evaluated
     * by the library on the host at emission).  to_string_timestamp64 shares
     * to_string_uint64's call pointer: whichever entry matches first yields the same
     * bytes (both print the decimal value) */
    {&ex::to_string_nil, EVQL_FN(EVQL_FAM_TO_STRING, EVQL_TS_NIL), "to_string#string/nil;"},
    {&ex::to_string_uint64, EVQL_FN(EVQL_FAM_TO_STRING, EVQL_TS_UINT64), "to_string#string/uint64;"},
    {&ex::to_string_int64, EVQL_FN(EVQL_FAM_TO_STRING, EVQL_TS_INT64), "to_string#string/int64;"},
    {&ex::to_string_float64, EVQL_FN(EVQL_FAM_TO_STRING, EVQL_TS_FLOAT64),
     "to_string#string/float64;"},
    {&ex::to_string_bool, EVQL_FN(EVQL_FAM_TO_STRING, EVQL_TS_BOOL), "to_string#string/bool;"},
    {&ex::concat, EVQL_FN(EVQL_FAM_CONCAT, EVQL_TS_STRING), "concat#string/string;string;"},
    {&ex::lcase, EVQL_FN(EVQL_FAM_LCASE, EVQL_TS_STRING), "lcase#string/string;"},
    {&ex::ucase, EVQL_FN(EVQL_FAM_UCASE, EVQL_TS_STRING), "ucase#string/string;"},
    {&ex::substring, EVQL_FN(EVQL_FAM_SUBSTRING, EVQL_TS_STRING), "substring#string/string;int64;"},
    {&ex::ltrim, EVQL_FN(EVQL_FAM_LTRIM, EVQL_TS_STRING), "ltrim#string/string;"},
    {&ex::rtrim, EVQL_FN(EVQL_FAM_RTRIM, EVQL_TS_STRING), "rtrim#string/string;"},
    {&ex::startswith, EVQL_FN(EVQL_FAM_STARTSWITH, EVQL_TS_STRING),
     "startswith#bool/string;string;"},
    {&ex::endswith, EVQL_FN(EVQL_FAM_ENDSWITH, EVQL_TS_STRING), "endswith#bool/string;string;"},
};

struct AggEntry {
  const csql::SFunction* fn;
  uint32_t id;
  const char* symbol;
};

/* sql/expressions/aggregate.h (aggregate.cc:35-219): the only aggregates this
 * snapshot of the reference registers (defaults.cc:49-54) */
const AggEntry kAgg[] = {
    {&ex::count, EVQL_AGG_COUNT, "count#uint64/nil;"},
    {&ex::sum_uint64, EVQL_AGG_SUM_UINT64, "sum#uint64/uint64;"},
    {&ex::sum_int64, EVQL_AGG_SUM_INT64, "sum#int64/int64;"},
    {&ex::count_distinct_uint64, EVQL_AGG_COUNT_DISTINCT_UINT64, "count_distinct#uint64/uint64;"},
};

const PureEntry* findPure(intptr_t call) {
  for (const auto& e : kPure) {
    if ((intptr_t) e.fn->vtable.call == call) return &e;
  }
  return nullptr;
}

}  // namespace

std::string pureFunctionSymbol(int64_t fn_id) {
  for (const auto& e : kPure) {
    if (e.id == fn_id) return e.symbol;
  }
  return "";
}

std::string aggregateSymbol(uint32_t aggregate_fn) {
  for (const auto& e : kAgg) {
    if (e.id == aggregate_fn) return e.symbol;
  }
  return "";
}

void LoweredProgram::seal() {
  c.code = code.data();
  c.code_len = (uint32_t) code.size();
  c.static_storage = literals.data();
  c.static_storage_len = literals.size();
}

bool lowerProgram(const csql::vm::Program* p, LoweredProgram* out, std::string* why) {
  out->code.clear();
  out->literals.clear();
  out->symbols.clear();
  const AggEntry* agg = nullptr;
  if (p->method_accumulate.offset > 0) {
    /* Compiler::compile copies the aggregate's vtable into the program
     * (compiler.cc:80-86); merge identifies the function */
    for (const auto& e : kAgg) {
      if (p->instance_merge == e.fn->vtable.merge &&
          p->instance_savestate == e.fn->vtable.savestate) {
        agg = &e;
        break;
      }
    }
    if (!agg) {
      if (why) *why = "aggregate function outside the lowerable set";
      return false;
    }
  }

  for (const auto& op : p->instructions) {
    evql_instr_t i;
    i.op = (uint32_t) op.type;
    i.argt = (uint32_t) op.argt;
    i.arg0 = (int64_t) op.arg0;
    std::string sym;
    switch (op.type) {
      case csql::vm::X_CALL_PURE: {
        const PureEntry* e = findPure(op.arg0);
        if (!e) {
          if (why) *why = "pure function outside the lowerable set";
          return false;
        }
        i.arg0 = e->id;
        i.argt = 0;
        sym = e->symbol;
        break;
      }
      case csql::vm::X_CALL_INSTANCE:
        i.argt = 0;
        if (agg && op.arg0 == (intptr_t) agg->fn->vtable.accumulate) {
          i.arg0 = EVQL_INSTANCE_ACCUMULATE;
        } else if (agg && op.arg0 == (intptr_t) agg->fn->vtable.get) {
          i.arg0 = EVQL_INSTANCE_GET;
        } else {
          /* a second, different aggregate in one expression: the reference runs
           * its get() on the first aggregate's instance (compiler.cc:67-100,
           * SURVEY "one aggregate call per select-list expression").  Only the
           * same-function case is well defined; refuse the rest */
          if (why) *why = "instance call that does not belong to the program's aggregate";
          return false;
        }
        sym = agg->symbol;
        break;
      case csql::vm::X_LITERAL: {
        const void* lit = (const void*) op.arg0;
        size_t len = csql::sql_sizeof(op.argt, lit); /* value bytes + tag */
        i.arg0 = (int64_t) out->literals.size();
        out->literals.insert(out->literals.end(), (const uint8_t*) lit,
                             (const uint8_t*) lit + len);
        break;
      }
      case csql::vm::X_INPUT:
        break;
      case csql::vm::X_JUMP:
      case csql::vm::X_CJUMP:
      case csql::vm::X_RETURN:
        i.argt = 0;
        break;
    }
    out->code.push_back(i);
    out->symbols.push_back(sym);
  }

  out->c.method_call = (uint32_t) p->method_call.offset;
  out->c.method_accumulate = (uint32_t) p->method_accumulate.offset;
  out->c.return_type = (uint32_t) p->return_type;
  out->c.aggregate_fn = agg ? agg->id : (uint32_t) EVQL_AGG_NONE;
  out->seal();
  return true;
}

}  // namespace evql_adapter
