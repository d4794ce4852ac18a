#include "plan_ir.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <sstream>

namespace evql {

namespace {

uint32_t slot_to_type(int ts) {
  switch (ts) {
    case EVQL_TS_UINT64: return EVQL_T_UINT64;
    case EVQL_TS_INT64: return EVQL_T_INT64;
    case EVQL_TS_FLOAT64: return EVQL_T_FLOAT64;
    case EVQL_TS_BOOL: return EVQL_T_BOOL;
    case EVQL_TS_STRING: return EVQL_T_STRING;
    case EVQL_TS_TIMESTAMP64: return EVQL_T_TIMESTAMP64;
    default: return EVQL_T_NIL;
  }
}

int family_arity(int fam) {
  switch (fam) {
    case EVQL_FAM_NEG:
    case EVQL_FAM_TO_NIL:
    case EVQL_FAM_TO_INT64:
    case EVQL_FAM_TO_TIMESTAMP64:
    case EVQL_FAM_TO_STRING:
    case EVQL_FAM_LCASE:
    case EVQL_FAM_UCASE:
    case EVQL_FAM_LTRIM:
    case EVQL_FAM_RTRIM:
      return 1;
    default:
      return 2;
  }
}

uint32_t call_return_type(int fam, int ts) {
  switch (fam) {
    case EVQL_FAM_LOGICAL_AND:
    case EVQL_FAM_LOGICAL_OR:
    case EVQL_FAM_NEG:
    case EVQL_FAM_EQ:
    case EVQL_FAM_NEQ:
    case EVQL_FAM_LT:
    case EVQL_FAM_LTE:
    case EVQL_FAM_GT:
    case EVQL_FAM_GTE:
    case EVQL_FAM_STARTSWITH:
    case EVQL_FAM_ENDSWITH:
      return EVQL_T_BOOL;
    case EVQL_FAM_TO_STRING:
    case EVQL_FAM_CONCAT:
    case EVQL_FAM_LCASE:
    case EVQL_FAM_UCASE:
    case EVQL_FAM_SUBSTRING:
    case EVQL_FAM_LTRIM:
    case EVQL_FAM_RTRIM:
      return EVQL_T_STRING;
    case EVQL_FAM_CMP:
    case EVQL_FAM_TO_INT64:
      return EVQL_T_INT64;
    case EVQL_FAM_TO_NIL:
      return EVQL_T_NIL;
    case EVQL_FAM_TO_TIMESTAMP64:
      return EVQL_T_TIMESTAMP64;
    default:
      return slot_to_type(ts);
  }
}

struct Decompiler {
  const evql_program_t& p;
  std::string err;
  bool unsupported = false;

  explicit Decompiler(const evql_program_t& prog) : p(prog) {}

  bool fail(const std::string& m, bool unsup = false) {
    if (err.empty()) err = m;
    unsupported = unsupported || unsup;
    return false;
  }

  ExprPtr literal(const evql_instr_t& op) {
    auto e = std::make_shared<Expr>();
    e->kind = Expr::LITERAL;
    e->type = op.argt;
    size_t off = size_t(op.arg0);
    const uint8_t* s = p.static_storage;
    size_t n = p.static_storage_len;
    switch (op.argt) {
      case EVQL_T_UINT64:
      case EVQL_T_INT64:
      case EVQL_T_FLOAT64:
      case EVQL_T_TIMESTAMP64:
        if (off + 9 > n) return nullptr;
        memcpy(&e->lit_bits, s + off, 8);
        e->lit_tag = s[off + 8];
        break;
      case EVQL_T_BOOL:
        if (off + 2 > n) return nullptr;
        e->lit_bits = s[off] ? 1 : 0;
        e->lit_tag = s[off + 1];
        break;
      case EVQL_T_STRING: {
        if (off + 5 > n) return nullptr;
        uint32_t l;
        memcpy(&l, s + off, 4);
        if (off + 4 + size_t(l) + 1 > n) return nullptr;
        e->lit_str.assign(reinterpret_cast<const char*>(s + off + 4), l);
        e->lit_tag = s[off + 4 + l];
        break;
      }
      default:
        return nullptr;
    }
    return e;
  }

  // simulates [pc, ...) until X_RETURN / end / an X_CALL_INSTANCE accumulate;
  // `stop` = exclusive upper bound for structured sub-ranges
  bool run(uint32_t pc, uint32_t stop, std::vector<ExprPtr>* st, bool* saw_acc) {
    while (pc < stop) {
      if (pc >= p.code_len) return fail("program counter out of range");
      const evql_instr_t& op = p.code[pc];
      switch (op.op) {
        case EVQL_X_INPUT: {
          if (op.argt == EVQL_T_NIL) return fail("NIL input", true);
          auto e = std::make_shared<Expr>();
          e->kind = Expr::INPUT;
          e->type = op.argt;
          e->input = uint32_t(op.arg0);
          st->push_back(e);
          ++pc;
          break;
        }
        case EVQL_X_LITERAL: {
          auto e = literal(op);
          if (!e) return fail("bad literal");
          st->push_back(e);
          ++pc;
          break;
        }
        case EVQL_X_CALL_PURE: {
          int fam = int(op.arg0 / 16), ts = int(op.arg0 % 16);
          if (fam < EVQL_FAM_LOGICAL_AND || fam > EVQL_FAM_LAST) {
            return fail("function not lowerable", true);
          }
          int ar = family_arity(fam);
          if (int(st->size()) < ar) return fail("stack underflow");
          auto e = std::make_shared<Expr>();
          e->kind = Expr::CALL;
          e->family = fam;
          e->type_slot = ts;
          e->type = call_return_type(fam, ts);
          e->args.assign(st->end() - ar, st->end());
          st->resize(st->size() - ar);
          st->push_back(e);
          ++pc;
          break;
        }
        case EVQL_X_CALL_INSTANCE: {
          if (op.arg0 == EVQL_INSTANCE_GET) {
            auto e = std::make_shared<Expr>();
            e->kind = Expr::AGG_GET;
            switch (p.aggregate_fn) {
              case EVQL_AGG_COUNT:
              case EVQL_AGG_SUM_UINT64:
              case EVQL_AGG_MIN_UINT64:
              case EVQL_AGG_MAX_UINT64:
                e->type = EVQL_T_UINT64;
                break;
              case EVQL_AGG_SUM_INT64:
              case EVQL_AGG_MIN_INT64:
              case EVQL_AGG_MAX_INT64:
                e->type = EVQL_T_INT64;
                break;
              case EVQL_AGG_COUNT_DISTINCT_UINT64:
                e->type = EVQL_T_UINT64;
                break;
              case EVQL_AGG_SUM_FLOAT64:
              case EVQL_AGG_MIN_FLOAT64:
              case EVQL_AGG_MAX_FLOAT64:
              case EVQL_AGG_MEAN_UINT64:
              case EVQL_AGG_MEAN_INT64:
              case EVQL_AGG_MEAN_FLOAT64:
                e->type = EVQL_T_FLOAT64;
                break;
              default:
                return fail("aggregate not lowerable", true);
            }
            st->push_back(e);
            ++pc;
            break;
          }
          if (op.arg0 == EVQL_INSTANCE_ACCUMULATE) {
            *saw_acc = true;
            return true;
          }
          return fail("bad X_CALL_INSTANCE");
        }
        case EVQL_X_CJUMP: {
          // IF(c,t,f): c; CJUMP ->T; f; JUMP ->E; T: t    (compiler.cc:174-209)
          if (st->empty()) return fail("stack underflow");
          uint32_t T = uint32_t(op.arg0);
          if (T == 0 || T > p.code_len || T <= pc + 1) {
            return fail("unstructured jump", true);
          }
          const evql_instr_t& j = p.code[T - 1];
          if (j.op != EVQL_X_JUMP) return fail("unstructured jump", true);
          uint32_t E = uint32_t(j.arg0);
          if (E < T || E > p.code_len) return fail("unstructured jump", true);
          auto e = std::make_shared<Expr>();
          e->kind = Expr::IF;
          ExprPtr cond = st->back();
          st->pop_back();
          std::vector<ExprPtr> fs, ts;
          bool dummy = false;
          if (!run(pc + 1, T - 1, &fs, &dummy)) return false;
          if (!run(T, E, &ts, &dummy)) return false;
          if (fs.size() != 1 || ts.size() != 1) return fail("malformed IF");
          e->args = {cond, ts[0], fs[0]};
          e->type = ts[0]->type;
          st->push_back(e);
          pc = E;
          break;
        }
        case EVQL_X_JUMP:
          return fail("unstructured jump", true);
        case EVQL_X_RETURN:
          return true;
        default:
          return fail("bad opcode");
      }
    }
    return true;
  }
};

}  // namespace

std::string lower_program(const evql_program_t& p, LoweredProgram* out,
                          bool* unsupported) {
  *unsupported = false;
  Decompiler d(p);
  out->return_type = p.return_type;
  out->is_aggregate = p.method_accumulate > 0;
  out->aggregate_fn = out->is_aggregate ? p.aggregate_fn : uint32_t(EVQL_AGG_NONE);
  if (out->is_aggregate && (p.aggregate_fn == EVQL_AGG_NONE ||
                            p.aggregate_fn > EVQL_AGG_COUNT_DISTINCT_UINT64)) {
    *unsupported = true;
    return "aggregate function not lowerable";
  }
  std::vector<ExprPtr> st;
  bool saw_acc = false;
  if (!d.run(p.method_call, p.code_len, &st, &saw_acc)) {
    *unsupported = d.unsupported;
    return d.err;
  }
  if (p.return_type == EVQL_T_NIL) {
    // e.g. to_nil(x) as a select expression: nothing observable
    if (!st.empty()) out->call = st.back();
  } else {
    if (st.size() != 1) return "malformed program: stack depth != 1";
    out->call = st[0];
  }
  if (out->is_aggregate) {
    std::vector<ExprPtr> as;
    saw_acc = false;
    if (!d.run(p.method_accumulate, p.code_len, &as, &saw_acc)) {
      *unsupported = d.unsupported;
      return d.err;
    }
    if (!saw_acc) return "malformed aggregate program";
    out->acc_args = as;
  }
  return std::string();
}

bool expr_equal(const ExprPtr& a, const ExprPtr& b) {
  if (!a || !b) return a == b;
  return expr_fingerprint(a) == expr_fingerprint(b);
}

void expr_inputs(const ExprPtr& e, std::vector<uint32_t>* out) {
  if (!e) return;
  if (e->kind == Expr::INPUT) {
    for (auto i : *out) {
      if (i == e->input) return;
    }
    out->push_back(e->input);
  }
  for (const auto& a : e->args) expr_inputs(a, out);
}

bool expr_uses_strings(const ExprPtr& e) {
  if (!e) return false;
  if (e->type == EVQL_T_STRING) return true;
  for (const auto& a : e->args) {
    if (expr_uses_strings(a)) return true;
  }
  return false;
}

std::string expr_fingerprint(const ExprPtr& e) {
  if (!e) return "~";
  std::ostringstream s;
  switch (e->kind) {
    case Expr::INPUT:
      s << "in" << e->input << ":" << e->type;
      break;
    case Expr::LITERAL:
      s << "lit" << e->type << ":" << e->lit_bits << ":" << int(e->lit_tag) << ":"
        << e->lit_str.size() << ":" << e->lit_str;
      break;
    case Expr::CALL:
      s << "f" << e->family << "." << e->type_slot << "(";
      for (const auto& a : e->args) s << expr_fingerprint(a) << ",";
      s << ")";
      break;
    case Expr::IF:
      s << "if(" << expr_fingerprint(e->args[0]) << "," << expr_fingerprint(e->args[1])
        << "," << expr_fingerprint(e->args[2]) << ")";
      break;
    case Expr::AGG_GET:
      s << "agg";
      break;
  }
  return s.str();
}

// ---------------------------------------------------------------------------
// host scalar evaluation
// ---------------------------------------------------------------------------
namespace {
double as_f64(uint64_t b) {
  double d;
  memcpy(&d, &b, 8);
  return d;
}
uint64_t f64_bits(double d) {
  uint64_t b;
  memcpy(&b, &d, 8);
  return b;
}

int str_cmp(const std::string& a, const std::string& b) {
  // boolean.cc:150-166: strncmp on the common prefix, then length
  size_t m = a.size() < b.size() ? a.size() : b.size();
  int c = m ? strncmp(a.data(), b.data(), m) : 0;
  if (c != 0) return c < 0 ? -1 : 1;
  if (a.size() < b.size()) return -1;
  if (a.size() > b.size()) return 1;
  return 0;
}
}  // namespace

std::string eval_expr(const ExprPtr& e, const std::vector<Value>& inputs,
                      const Value* agg, Value* out) {
  switch (e->kind) {
    case Expr::INPUT:
      if (e->input >= inputs.size()) return "invalid input index";
      *out = inputs[e->input];
      out->type = e->type;
      return "";
    case Expr::LITERAL:
      out->type = e->type;
      out->bits = e->lit_bits;
      out->str = e->lit_str;
      out->tag = e->lit_tag;
      return "";
    case Expr::AGG_GET:
      if (!agg) return "aggregate value unavailable";
      *out = *agg;
      return "";
    case Expr::IF: {
      Value c;
      std::string r = eval_expr(e->args[0], inputs, agg, &c);
      if (!r.empty()) return r;
      return eval_expr(c.bits ? e->args[1] : e->args[2], inputs, agg, out);
    }
    case Expr::CALL:
      break;
  }
  std::vector<Value> a(e->args.size());
  for (size_t i = 0; i < e->args.size(); ++i) {
    std::string r = eval_expr(e->args[i], inputs, agg, &a[i]);
    if (!r.empty()) return r;
  }
  out->type = e->type;
  out->tag = 0;  // pure functions always push tag 0
  out->str.clear();
  const int fam = e->family, ts = e->type_slot;
  switch (fam) {
    case EVQL_FAM_LOGICAL_AND:
      out->bits = (a[0].bits && a[1].bits) ? 1 : 0;
      return "";
    case EVQL_FAM_LOGICAL_OR:
      out->bits = (a[0].bits || a[1].bits) ? 1 : 0;
      return "";
    case EVQL_FAM_NEG:
      out->bits = a[0].bits ? 0 : 1;
      return "";
    case EVQL_FAM_CMP:
    case EVQL_FAM_EQ:
    case EVQL_FAM_NEQ:
    case EVQL_FAM_LT:
    case EVQL_FAM_LTE:
    case EVQL_FAM_GT:
    case EVQL_FAM_GTE: {
      int c;
      switch (ts) {
        case EVQL_TS_INT64: {
          int64_t l = int64_t(a[0].bits), r = int64_t(a[1].bits);
          c = l < r ? -1 : (l > r ? 1 : 0);
          break;
        }
        case EVQL_TS_FLOAT64: {
          double l = as_f64(a[0].bits), r = as_f64(a[1].bits);
          c = l < r ? -1 : (l > r ? 1 : (l == r ? 0 : 2));
          break;
        }
        case EVQL_TS_STRING:
          c = str_cmp(a[0].str, a[1].str);
          // eq / neq use memcmp (boolean.cc:237-251, 357-371)
          if (fam == EVQL_FAM_EQ || fam == EVQL_FAM_NEQ) c = a[0].str == a[1].str ? 0 : 1;
          break;
        default: {
          uint64_t l = a[0].bits, r = a[1].bits;
          c = l < r ? -1 : (l > r ? 1 : 0);
        }
      }
      switch (fam) {
        case EVQL_FAM_CMP: out->bits = uint64_t(int64_t(c == 2 ? 0 : c)); break;
        case EVQL_FAM_EQ: out->bits = c == 0; break;
        case EVQL_FAM_NEQ: out->bits = c != 0; break;
        case EVQL_FAM_LT: out->bits = c == -1; break;
        case EVQL_FAM_LTE: out->bits = (c == -1 || c == 0); break;
        case EVQL_FAM_GT: out->bits = c == 1; break;
        case EVQL_FAM_GTE: out->bits = (c == 1 || c == 0); break;
      }
      return "";
    }
    case EVQL_FAM_ADD:
    case EVQL_FAM_SUB:
    case EVQL_FAM_MUL:
    case EVQL_FAM_DIV:
    case EVQL_FAM_MOD:
    case EVQL_FAM_POW:
      if (ts == EVQL_TS_FLOAT64) {
        double l = as_f64(a[0].bits), r = as_f64(a[1].bits), o = 0;
        switch (fam) {
          case EVQL_FAM_ADD: o = l + r; break;
          case EVQL_FAM_SUB: o = l - r; break;
          case EVQL_FAM_MUL: o = l * r; break;
          case EVQL_FAM_DIV: o = l / r; break;
          case EVQL_FAM_MOD: o = fmod(l, r); break;
          case EVQL_FAM_POW: o = pow(l, r); break;
        }
        out->bits = f64_bits(o);
        return "";
      }
      if (ts == EVQL_TS_UINT64) {
        uint64_t l = a[0].bits, r = a[1].bits, o = 0;
        switch (fam) {
          case EVQL_FAM_ADD: o = l + r; break;
          case EVQL_FAM_SUB: o = l - r; break;
          case EVQL_FAM_MUL: o = l * r; break;
          case EVQL_FAM_DIV:
            if (r == 0) return "division by zero";
            o = l / r;
            break;
          case EVQL_FAM_MOD:
            if (r == 0) return "modulo by zero";
            o = l % r;
            break;
          case EVQL_FAM_POW: o = uint64_t(pow(double(l), double(r))); break;
        }
        out->bits = o;
        return "";
      }
      if (ts == EVQL_TS_INT64) {
        int64_t l = int64_t(a[0].bits), r = int64_t(a[1].bits), o = 0;
        switch (fam) {
          case EVQL_FAM_ADD: o = int64_t(uint64_t(l) + uint64_t(r)); break;
          case EVQL_FAM_SUB: o = int64_t(uint64_t(l) - uint64_t(r)); break;
          case EVQL_FAM_MUL: o = int64_t(uint64_t(l) * uint64_t(r)); break;
          case EVQL_FAM_DIV:
            if (r == 0) return "division by zero";
            o = (l == INT64_MIN && r == -1) ? INT64_MIN : l / r;
            break;
          case EVQL_FAM_MOD:
            if (r == 0) return "modulo by zero";
            o = (r == -1) ? 0 : l % r;
            break;
          case EVQL_FAM_POW: o = int64_t(pow(double(l), double(r))); break;
        }
        out->bits = uint64_t(o);
        return "";
      }
      return "type error in arithmetic";
    case EVQL_FAM_TO_NIL:
      out->bits = 0;
      return "";
    case EVQL_FAM_TO_INT64:
      if (ts == EVQL_TS_FLOAT64) out->bits = uint64_t(int64_t(as_f64(a[0].bits)));
      else out->bits = a[0].bits;
      return "";
    case EVQL_FAM_TO_TIMESTAMP64:
      if (ts == EVQL_TS_FLOAT64) out->bits = uint64_t(as_f64(a[0].bits));
      else out->bits = a[0].bits;
      return "";
    // ---- strings (expressions/string.cc, conversion.cc:140-215) -----------------------
    case EVQL_FAM_TO_STRING:
      // sql_tostring (svalue.cc:592-660): a NULL tag reads "NULL"; std::to_string for the
      // numbers (doubles: "%f"); timestamps go through to_string_uint64_call: decimal
      if (a[0].tag & EVQL_STAG_NULL) {
        out->str = "NULL";
        return "";
      }
      switch (ts) {
        case EVQL_TS_INT64: out->str = std::to_string((long long) int64_t(a[0].bits)); break;
        case EVQL_TS_FLOAT64: out->str = std::to_string(as_f64(a[0].bits)); break;
        case EVQL_TS_BOOL: out->str = a[0].bits ? "true" : "false"; break;
        case EVQL_TS_STRING: out->str = a[0].str; break;
        case EVQL_TS_NIL: out->str = "NULL"; break;
        default: out->str = std::to_string((unsigned long long) a[0].bits);
      }
      return "";
    case EVQL_FAM_CONCAT:
      out->str = a[0].str + a[1].str;
      return "";
    case EVQL_FAM_LCASE:
    case EVQL_FAM_UCASE:
      out->str = a[0].str;
      for (char& ch : out->str) {  // std::tolower / toupper in the "C" locale
        if (fam == EVQL_FAM_LCASE && ch >= 'A' && ch <= 'Z') ch = char(ch - 'A' + 'a');
        if (fam == EVQL_FAM_UCASE && ch >= 'a' && ch <= 'z') ch = char(ch - 'a' + 'A');
      }
      return "";
    case EVQL_FAM_LTRIM: {
      size_t i = 0;
      while (i < a[0].str.size() && a[0].str[i] == ' ') ++i;
      out->str = a[0].str.substr(i);
      return "";
    }
    case EVQL_FAM_RTRIM: {
      size_t n = a[0].str.size();
      while (n > 0 && a[0].str[n - 1] == ' ') --n;
      out->str = a[0].str.substr(0, n);
      return "";
    }
    case EVQL_FAM_SUBSTRING: {  // string.cc substring_call
      int64_t cur = int64_t(a[1].bits);
      const int64_t len = int64_t(a[0].str.size());
      if (cur == 0 || len == 0) return "";
      if (cur < 0) {
        cur += len;
        if (cur < 0) return "";
      } else {
        cur = std::min(cur - 1, len - 1);
      }
      out->str = a[0].str.substr(size_t(cur));
      return "";
    }
    case EVQL_FAM_STARTSWITH:
      out->bits = a[0].str.size() >= a[1].str.size() &&
                  a[0].str.compare(0, a[1].str.size(), a[1].str) == 0;
      return "";
    case EVQL_FAM_ENDSWITH:
      out->bits = a[0].str.size() >= a[1].str.size() &&
                  a[0].str.compare(a[0].str.size() - a[1].str.size(), a[1].str.size(), a[1].str) == 0;
      return "";
  }
  return "function not lowerable";
}

void append_svector(uint32_t type, const Value& v, std::vector<uint8_t>* out) {
  switch (type) {
    case EVQL_T_NIL:
      return;  // popVector on NIL appends nothing (svalue.cc:793-794)
    case EVQL_T_BOOL:
      out->push_back(v.bits ? 1 : 0);
      out->push_back(v.tag);
      return;
    case EVQL_T_STRING: {
      uint32_t l = uint32_t(v.str.size());
      const uint8_t* lp = reinterpret_cast<const uint8_t*>(&l);
      out->insert(out->end(), lp, lp + 4);
      out->insert(out->end(), v.str.begin(), v.str.end());
      out->push_back(v.tag);
      return;
    }
    default: {
      const uint8_t* p = reinterpret_cast<const uint8_t*>(&v.bits);
      out->insert(out->end(), p, p + 8);
      out->push_back(v.tag);
    }
  }
}

}  // namespace evql
