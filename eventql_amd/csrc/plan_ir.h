// plan_ir.h -- lowering of csql bytecode (vm::Program mirrors, see
// include/evql_gpu.h) into expression trees, plus a scalar evaluator used at
// result-emission time (GroupByExpression::nextBatch, groupby.cc:187-220:
// `method_call` of every select expression is run once per group).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include "../../include/evql_gpu.h"

namespace evql {

struct Expr;
using ExprPtr = std::shared_ptr<Expr>;

struct Expr {
  enum Kind { INPUT, LITERAL, CALL, IF, AGG_GET } kind;
  uint32_t type = EVQL_T_NIL;  // evql_stype of the value produced
  // INPUT
  uint32_t input = 0;
  // LITERAL: value bits (u64/i64/f64/bool) or string bytes; tag
  uint64_t lit_bits = 0;
  std::string lit_str;
  uint8_t lit_tag = 0;
  // CALL
  int family = 0;
  int type_slot = 0;
  // children: CALL args (left to right); IF: cond, true, false
  std::vector<ExprPtr> args;
};

// one lowered vm::Program
struct LoweredProgram {
  ExprPtr call;  // method_call expression (may contain AGG_GET)
  bool is_aggregate = false;
  uint32_t aggregate_fn = EVQL_AGG_NONE;
  std::vector<ExprPtr> acc_args;  // arguments pushed before `accumulate`
  uint32_t return_type = EVQL_T_NIL;
};

// Decompiles the post-order stack code.  Returns "" or an error message; sets
// *unsupported when the program uses something outside the lowerable op table
//  so that the caller can answer EVQL_ENOTSUP.
std::string lower_program(const evql_program_t& p, LoweredProgram* out,
                          bool* unsupported);

// structural helpers
bool expr_equal(const ExprPtr& a, const ExprPtr& b);
void expr_inputs(const ExprPtr& e, std::vector<uint32_t>* out);
bool expr_uses_strings(const ExprPtr& e);
std::string expr_fingerprint(const ExprPtr& e);

// ---------------------------------------------------------------------------
// scalar values and evaluation on the host (result emission only)
// ---------------------------------------------------------------------------
struct Value {
  uint32_t type = EVQL_T_NIL;
  uint64_t bits = 0;  // u64 / i64 / f64 bits / bool
  std::string str;
  uint8_t tag = 0;
};

// Evaluates `e`; inputs[i] is the value of X_INPUT(i); agg is the value pushed
// by X_CALL_INSTANCE get.  Returns "" or an error message ("division by zero").
std::string eval_expr(const ExprPtr& e, const std::vector<Value>& inputs,
                      const Value* agg, Value* out);

// appends the packed SVector element (svalue.cc:410-517) of v, typed `type`
void append_svector(uint32_t type, const Value& v, std::vector<uint8_t>* out);

}  // namespace evql
