#include "codegen.h"
#include <cstdio>
#include <sstream>

namespace evql {

const char* device_library_source() {
  static const char* src =
#include "evql_device.inc"
      ;
  return src;
}

namespace {

const char* ctype(uint32_t t) {
  switch (t) {
    case EVQL_T_INT64: return "i64";
    case EVQL_T_FLOAT64: return "double";
    case EVQL_T_BOOL: return "bool";
    default: return "u64";
  }
}

std::string hex64(uint64_t v) {
  char b[32];
  snprintf(b, sizeof(b), "0x%016llxull", (unsigned long long) v);
  return b;
}

const char* op_name(int op) {
  static const char* n[] = {"EVQL_OP_ADD_U64", "EVQL_OP_ADD_F64", "EVQL_OP_MIN_U64",
                            "EVQL_OP_MAX_U64", "EVQL_OP_MIN_I64", "EVQL_OP_MAX_I64",
                            "EVQL_OP_MIN_F64", "EVQL_OP_MAX_F64"};
  return n[op];
}

struct Val {
  std::string v;  // payload expression (already typed)
  std::string g;  // tag expression (u32)
  uint32_t type;
};

struct Emitter {
  std::ostringstream o;
  int tmp = 0;
  std::string ind = "  ";

  std::string fresh(const char* p) {
    char b[32];
    snprintf(b, sizeof(b), "%s%d", p, tmp++);
    return b;
  }

  // string literals referenced by compares; declared at file scope by the caller
  std::vector<std::string> string_literals;

  std::string str_operand(const ExprPtr& x) {
    char b[96];
    if (x->kind == Expr::INPUT) {
      snprintf(b, sizeof(b), "evql_col_str(A, A.col[%u], row)", x->input);
      return b;
    }
    snprintf(b, sizeof(b), "evql_lit_str(evql_slit%zu, %zuu)", string_literals.size(),
             x->lit_str.size());
    string_literals.push_back(x->lit_str);
    return b;
  }

  // payload reinterpreted as raw 64 bits
  static std::string as_bits(const Val& x) {
    switch (x.type) {
      case EVQL_T_FLOAT64: return "evql_f64_bits(" + x.v + ")";
      case EVQL_T_BOOL: return "((u64) (" + x.v + " ? 1 : 0))";
      default: return "((u64) " + x.v + ")";
    }
  }

  Val emit(const ExprPtr& e) {
    switch (e->kind) {
      case Expr::INPUT: {
        char v[16], g[16];
        snprintf(v, sizeof(v), "c%u", e->input);
        snprintf(g, sizeof(g), "g%u", e->input);
        return {v, g, e->type};
      }
      case Expr::LITERAL: {
        std::string v;
        switch (e->type) {
          case EVQL_T_FLOAT64: v = "evql_as_f64(" + hex64(e->lit_bits) + ")"; break;
          case EVQL_T_INT64: v = "((i64) " + hex64(e->lit_bits) + ")"; break;
          case EVQL_T_BOOL: v = e->lit_bits ? "true" : "false"; break;
          default: v = hex64(e->lit_bits);
        }
        return {v, e->lit_tag ? "1u" : "0u", e->type};
      }
      case Expr::AGG_GET:
        return {"0", "0u", e->type};  // never reached on the device
      case Expr::IF: {
        Val c = emit(e->args[0]);
        std::string t = fresh("t"), g = fresh("q");
        o << ind << ctype(e->type) << " " << t << "; u32 " << g << ";\n";
        o << ind << "if (" << c.v << ") {\n";
        std::string save = ind;
        ind += "  ";
        Val a = emit(e->args[1]);
        o << ind << t << " = " << a.v << "; " << g << " = " << a.g << ";\n";
        ind = save;
        o << ind << "} else {\n";
        ind += "  ";
        Val b = emit(e->args[2]);
        o << ind << t << " = " << b.v << "; " << g << " = " << b.g << ";\n";
        ind = save;
        o << ind << "}\n";
        return {t, g, e->type};
      }
      case Expr::CALL:
        break;
    }
    const int fam = e->family, ts = e->type_slot;
    if (ts == EVQL_TS_STRING && ((fam >= EVQL_FAM_CMP && fam <= EVQL_FAM_GTE) ||
                                 fam == EVQL_FAM_STARTSWITH || fam == EVQL_FAM_ENDSWITH)) {
      // operands are string columns / literals (planner.cc strings_lowerable)
      const std::string l = str_operand(e->args[0]), r = str_operand(e->args[1]);
      const std::string c = "evql_str_cmp(" + l + ", " + r + ")";
      std::string rhs;
      switch (fam) {
        case EVQL_FAM_EQ: rhs = "evql_str_eq(" + l + ", " + r + ")"; break;
        case EVQL_FAM_NEQ: rhs = "(!evql_str_eq(" + l + ", " + r + "))"; break;
        case EVQL_FAM_LT: rhs = "(" + c + " < 0)"; break;
        case EVQL_FAM_LTE: rhs = "(" + c + " <= 0)"; break;
        case EVQL_FAM_GT: rhs = "(" + c + " > 0)"; break;
        case EVQL_FAM_GTE: rhs = "(" + c + " >= 0)"; break;
        case EVQL_FAM_STARTSWITH: rhs = "evql_str_affix(" + l + ", " + r + ", false)"; break;
        case EVQL_FAM_ENDSWITH: rhs = "evql_str_affix(" + l + ", " + r + ", true)"; break;
        default: rhs = "((i64) " + c + ")";
      }
      std::string t = fresh("t");
      o << ind << "const " << ctype(e->type) << " " << t << " = " << rhs << ";\n";
      return {t, "0u", e->type};
    }
    std::vector<Val> a;
    for (const auto& x : e->args) a.push_back(emit(x));
    std::string t = fresh("t");
    std::string rhs;
    auto bin = [&](const char* opr) { return "(" + a[0].v + " " + opr + " " + a[1].v + ")"; };
    switch (fam) {
      case EVQL_FAM_LOGICAL_AND: rhs = "(" + a[0].v + " & " + a[1].v + ")"; break;  // eager
      case EVQL_FAM_LOGICAL_OR: rhs = "(" + a[0].v + " | " + a[1].v + ")"; break;
      case EVQL_FAM_NEG: rhs = "(!" + a[0].v + ")"; break;
      case EVQL_FAM_EQ: rhs = bin("=="); break;
      case EVQL_FAM_NEQ: rhs = bin("!="); break;
      case EVQL_FAM_LT: rhs = bin("<"); break;
      case EVQL_FAM_LTE: rhs = bin("<="); break;
      case EVQL_FAM_GT: rhs = bin(">"); break;
      case EVQL_FAM_GTE: rhs = bin(">="); break;
      case EVQL_FAM_CMP:
        rhs = "((i64) (" + a[0].v + " < " + a[1].v + " ? -1 : (" + a[0].v + " > " + a[1].v +
              " ? 1 : 0)))";
        break;
      case EVQL_FAM_ADD: rhs = bin("+"); break;
      case EVQL_FAM_SUB: rhs = bin("-"); break;
      case EVQL_FAM_MUL: rhs = bin("*"); break;
      case EVQL_FAM_DIV:
      case EVQL_FAM_MOD: {
        const bool is_div = fam == EVQL_FAM_DIV;
        if (ts == EVQL_TS_FLOAT64) {
          rhs = is_div ? bin("/") : "fmod(" + a[0].v + ", " + a[1].v + ")";
        } else {
          // integer division by zero raises in the reference (math.cc:136-164):
          // flag it and keep going with 0; the host turns the flag into ERUNTIME
          o << ind << ctype(e->type) << " " << t << " = 0;\n";
          o << ind << "if (" << a[1].v << " == 0) { atomicOr(&A.status[0], "
            << (is_div ? "EVQL_ST_DIV_BY_ZERO" : "EVQL_ST_MOD_BY_ZERO") << "); } else {\n";
          if (ts == EVQL_TS_INT64) {
            if (is_div) {
              o << ind << "  " << t << " = (" << a[0].v
                << " == (i64) 0x8000000000000000ull && " << a[1].v << " == -1) ? " << a[0].v
                << " : " << a[0].v << " / " << a[1].v << ";\n";
            } else {
              o << ind << "  " << t << " = (" << a[1].v << " == -1) ? 0 : " << a[0].v << " % "
                << a[1].v << ";\n";
            }
          } else {
            o << ind << "  " << t << " = " << a[0].v << (is_div ? " / " : " % ") << a[1].v
              << ";\n";
          }
          o << ind << "}\n";
          return {t, "0u", e->type};
        }
        break;
      }
      case EVQL_FAM_POW:
        if (ts == EVQL_TS_FLOAT64) rhs = "pow(" + a[0].v + ", " + a[1].v + ")";
        else
          rhs = std::string("((") + ctype(e->type) + ") pow((double) " + a[0].v + ", (double) " +
                a[1].v + "))";
        break;
      case EVQL_FAM_TO_NIL: rhs = "0ull"; break;
      case EVQL_FAM_TO_INT64:
        rhs = "((i64) " + a[0].v + ")";
        break;
      case EVQL_FAM_TO_TIMESTAMP64:
        rhs = "((u64) " + a[0].v + ")";
        break;
      default: rhs = "0";
    }
    if (ts == EVQL_TS_INT64 && (fam == EVQL_FAM_ADD || fam == EVQL_FAM_SUB || fam == EVQL_FAM_MUL)) {
      // wrap-around like the reference's two's complement arithmetic, without UB
      const char* opr = fam == EVQL_FAM_ADD ? "+" : (fam == EVQL_FAM_SUB ? "-" : "*");
      rhs = "((i64) ((u64) " + a[0].v + " " + opr + " (u64) " + a[1].v + "))";
    }
    o << ind << "const " << ctype(e->type) << " " << t << " = " << rhs << ";\n";
    return {t, "0u", e->type};
  }
};

}  // namespace

// kernel skeletons (scan, partition count / scatter / aggregate)
#include "codegen_kernels.inc"

}  // namespace evql
