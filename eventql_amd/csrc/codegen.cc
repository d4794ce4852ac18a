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
    std::vector<Val> a;
    for (const auto& x : e->args) a.push_back(emit(x));
    std::string t = fresh("t");
    const int fam = e->family, ts = e->type_slot;
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

std::string generate_kernel_source(const KernelPlan& kp) {
  std::ostringstream s;
  const int NC = int(kp.cols.size());
  const int NW = int(kp.states.size());
  const int S = kp.lds_slots;
  const bool grouped = kp.key_mode != KEY_NONE;
  const int W = kp.words_per_slot();
  const int SB = kp.state_word_base();

  s << "// generated by eventql_amd codegen -- one fused scan/filter/GROUP BY kernel\n";
  s << "#define EVQL_BLOCK " << kp.block << "\n";
  s << "#define EVQL_UNROLL " << kp.unroll << "\n";
  s << "#define EVQL_TILE_ROWS " << kp.tile_rows() << "\n";
  s << "#define EVQL_LDS_SLOTS " << S << "\n";
  s << "#define EVQL_LSTRIDE " << (S + 2) << "\n";
  s << "#define EVQL_WORDS " << W << "\n";
  s << "#define EVQL_NSTATE " << NW << "\n";
  s << "\n";

  // update words of one row (count / sum / min / max state words), in order
  struct UpdWord {
    int word;
    int op;
  };
  std::vector<UpdWord> updw;
  for (const auto& a : kp.aggs) {
    switch (a.fn) {
      case EVQL_AGG_COUNT:
      case EVQL_AGG_SUM_UINT64:
      case EVQL_AGG_SUM_INT64:
        updw.push_back({a.first_word, 0});
        break;
      case EVQL_AGG_SUM_FLOAT64:
        updw.push_back({a.first_word, 1});
        break;
      case EVQL_AGG_MEAN_UINT64:
      case EVQL_AGG_MEAN_INT64:
      case EVQL_AGG_MEAN_FLOAT64:
        updw.push_back({a.first_word, 1});
        updw.push_back({a.first_word + 1, 0});
        break;
      default:
        updw.push_back({a.first_word, kp.states[a.first_word].op});
        updw.push_back({a.first_word + 1, 0});
    }
  }
  s << "struct EvqlAcc {\n  u64 passed;\n  u64 spilled;\n";
  if (!grouped) s << "  u64 w[" << (NW > 0 ? NW : 1) << "];\n";
  if (grouped && S > 0) {
    s << "  int rslot;\n  u64 rfirst;\n  u64 rw[" << (updw.empty() ? 1 : updw.size()) << "];\n";
  }
  s << "};\n\n";
  if (grouped && S > 0) {
    // fold the lane-private run accumulators into the LDS table (wave-converged)
    s << "__device__ __forceinline__ void evql_flush_private(u64* lds, EvqlAcc& acc) {\n";
    s << "  if (acc.rslot >= 0) {\n";
    if (kp.need_first_row) {
      s << "    { const u64 v = evql_wave_reduce<EVQL_OP_MIN_U64>(acc.rfirst);\n";
      s << "      if ((threadIdx.x & 63u) == 0) evql_atomic<EVQL_OP_MIN_U64>(&lds[" << kp.first_row_word()
        << " * EVQL_LSTRIDE + acc.rslot], v); }\n";
      s << "    acc.rfirst = 0xFFFFFFFFFFFFFFFFull;\n";
    }
    for (size_t i = 0; i < updw.size(); ++i) {
      s << "    { const u64 v = evql_wave_reduce<" << op_name(updw[i].op) << ">(acc.rw[" << i << "]);\n";
      s << "      if ((threadIdx.x & 63u) == 0) evql_atomic<" << op_name(updw[i].op) << ">(&lds["
        << (SB + updw[i].word) << " * EVQL_LSTRIDE + acc.rslot], v); }\n";
      s << "    acc.rw[" << i << "] = evql_op_identity<" << op_name(updw[i].op) << ">();\n";
    }
    s << "  }\n}\n\n";
  }

  // ---- per-row function ------------------------------------------------------
  s << "__device__ __forceinline__ void evql_row(const EvqlArgs& A, u64* lds, EvqlAcc& acc,\n"
       "                                         const bool bypass, const u64 row, const bool valid";
  for (int i = 0; i < NC; ++i) s << ", const u64 r" << i << ", const u32 g" << i;
  s << ") {\n";
  // The function is written without early returns: every lane of the wave
  // reaches the update section together, so that wave-level cooperation
  // (ballot / shuffle reductions over lanes that hit the same slot) is legal.
  s << "  bool live = valid;\n";
  if (kp.has_row_filter) {
    s << "  live = live && evql_row_filter(A.row_filter, A.row_filter_len, row);\n";
  }
  for (int i = 0; i < NC; ++i) {
    const ColAccess& c = kp.cols[i];
    s << "  const " << ctype(c.stype) << " c" << i << " = ";
    if (c.stype == EVQL_T_FLOAT64) {
      s << (c.from_uint_to_float ? "(double) r" : "evql_as_f64(r") << i
        << (c.from_uint_to_float ? "" : ")") << ";\n";
    } else if (c.stype == EVQL_T_BOOL) {
      s << "(r" << i << " != 0);\n";
    } else if (c.stype == EVQL_T_INT64) {
      s << "(i64) r" << i << ";\n";
    } else {
      s << "r" << i << ";\n";
    }
    s << "  (void) c" << i << "; (void) g" << i << ";\n";
  }
  Emitter em;
  em.ind = "    ";
  if (kp.where) {
    s << "  if (live) {\n";
    Val p = em.emit(kp.where);
    s << em.o.str();
    em.o.str("");
    s << "    live = " << p.v << ";\n  }\n";
  }
  s << "  acc.passed += live ? 1 : 0;\n";

  // aggregate arguments -> list of word updates
  struct Upd {
    int word;
    int op;
    std::string bits;
    std::string cond;  // "" = unconditional
  };
  std::vector<Upd> upd;
  const size_t nupd_words = [&] {
    size_t n = 0;
    for (const auto& a : kp.aggs) n += size_t(a.nwords);
    return n;
  }();
  s << "  u64 ident = 0; u64 ident2 = 0; bool knull = false;\n";
  for (size_t i = 0; i < nupd_words; ++i) {
    s << "  u64 ub" << i << " = 0; bool uc" << i << " = false;\n";
  }
  s << "  if (live) {\n";
  // group key -> ident / knull
  if (kp.key_mode == KEY_EXACT) {
    Val k = em.emit(kp.group[0]);
    s << em.o.str();
    em.o.str("");
    s << "    ident = " << Emitter::as_bits(k) << ";\n";
    s << "    knull = (" << k.g << " & 1u) != 0;\n";
  } else if (kp.key_mode == KEY_HASHED) {
    s << "    ident = 0x243f6a8885a308d3ull; ident2 = 0x13198a2e03707344ull;\n";
    for (const auto& g : kp.group) {
      Val k = em.emit(g);
      s << em.o.str();
      em.o.str("");
      s << "    ident = evql_hash_combine(ident, " << Emitter::as_bits(k) << ");\n";
      s << "    ident = evql_hash_combine(ident, (u64) (" << k.g << " & 1u));\n";
      s << "    ident2 = evql_mix64(ident2 * 0x9e3779b97f4a7c15ull + " << Emitter::as_bits(k)
        << ") ^ (u64) (" << k.g << " & 1u);\n";
    }
    // the all-ones pattern marks a free word
    s << "    if (ident == EVQL_EMPTY) ident = EVQL_EMPTY - 1;\n";
    s << "    if (ident2 == EVQL_EMPTY) ident2 = EVQL_EMPTY - 1;\n";
  }
  for (const auto& a : kp.aggs) {
    Val v{"0", "0u", EVQL_T_NIL};
    if (a.arg) {
      v = em.emit(a.arg);
      s << em.o.str();
      em.o.str("");
    }
    const std::string notnull = "((" + v.g + " & 1u) == 0)";
    const int w0 = a.first_word;
    auto push = [&](int word, int op, const std::string& bits, const std::string& cond) {
      const size_t i = upd.size();
      s << "    ub" << i << " = " << bits << "; uc" << i << " = " << (cond.empty() ? "true" : cond)
        << ";\n";
      char b[16], c[16];
      snprintf(b, sizeof(b), "ub%zu", i);
      snprintf(c, sizeof(c), "uc%zu", i);
      upd.push_back({word, op, b, c});
    };
    switch (a.fn) {
      case EVQL_AGG_COUNT:
        push(w0, 0, "1ull", "");
        break;
      case EVQL_AGG_SUM_UINT64:
      case EVQL_AGG_SUM_INT64:
        push(w0, 0, Emitter::as_bits(v), "");
        break;
      case EVQL_AGG_SUM_FLOAT64:
        push(w0, 1, Emitter::as_bits(v), "");
        break;
      case EVQL_AGG_MIN_UINT64:
      case EVQL_AGG_MAX_UINT64:
      case EVQL_AGG_MIN_INT64:
      case EVQL_AGG_MAX_INT64:
      case EVQL_AGG_MIN_FLOAT64:
      case EVQL_AGG_MAX_FLOAT64:
        push(w0, kp.states[w0].op, Emitter::as_bits(v), notnull);
        push(w0 + 1, 0, "1ull", notnull);
        break;
      case EVQL_AGG_MEAN_UINT64:
      case EVQL_AGG_MEAN_INT64:
      case EVQL_AGG_MEAN_FLOAT64: {
        std::string d = a.fn == EVQL_AGG_MEAN_FLOAT64 ? v.v : "((double) " + v.v + ")";
        push(w0, 1, "evql_f64_bits(" + d + ")", notnull);
        push(w0 + 1, 0, "1ull", notnull);
        break;
      }
      default:
        break;
    }
  }
  s << "  }\n";

  // address of word `w` of slot `slot`: LDS keeps word planes (bank spread),
  // the HBM table keeps the words of a slot adjacent (one line per group)
  auto word_at = [&](bool global, int w, const std::string& slot) {
    char b[160];
    if (global) {
      snprintf(b, sizeof(b), "A.gtab[(u64) %s * EVQL_WORDS + %d]", slot.c_str(), w);
    } else {
      snprintf(b, sizeof(b), "lds[%d * EVQL_LSTRIDE + %s]", w, slot.c_str());
    }
    return std::string(b);
  };
  auto emit_updates = [&](bool global, const char* slot, const char* indent) {
    if (kp.need_first_row) {
      s << indent << "evql_atomic<EVQL_OP_MIN_U64>(&" << word_at(global, kp.first_row_word(), slot)
        << ", row);\n";
    }
    for (const auto& u : upd) {
      s << indent << "if (" << u.cond << ") evql_atomic<" << op_name(u.op) << ">(&"
        << word_at(global, SB + u.word, slot) << ", " << u.bits << ");\n";
    }
  };

  if (!grouped) {
    for (const auto& u : upd) {
      s << "  if (" << u.cond << ") acc.w[" << u.word << "] = evql_combine<" << op_name(u.op)
        << ">(acc.w[" << u.word << "], " << u.bits << ");\n";
    }
  } else {
    if (S > 0) {
      // `bypass`: this workgroup has seen so many rows that found no LDS slot that
      // the table is evidently too small for the data (high cardinality with an
      // unknown hint): stop probing it, aggregate straight into the HBM table
      s << "  int s = -1;\n";
      s << "  if (live && !bypass) {\n";
      s << "    if (knull) { s = EVQL_LDS_SLOTS + 1; lds[s] = 0; }\n";
      s << "    else if (ident == EVQL_EMPTY) { s = EVQL_LDS_SLOTS; lds[s] = 0; }\n";
      if (kp.has_ident2()) {
        s << "    else s = evql_lds_find2<16>(lds, lds + EVQL_LSTRIDE, EVQL_LDS_SLOTS - 1, ident, ident2, "
             "(u32) evql_mix64(ident));\n";
      } else {
        s << "    else s = evql_lds_find<16>(lds, EVQL_LDS_SLOTS - 1, ident, (u32) evql_mix64(ident));\n";
      }
      s << "    if (s < 0) atomicAdd(reinterpret_cast<u32*>(&lds[EVQL_WORDS * EVQL_LSTRIDE]), 1u);\n";
      s << "  }\n";
      // Wave-uniform runs: when every pending lane of the wave hits the SAME slot
      // (one group, a dominant key, or input clustered by key) 64 LDS atomics on
      // one address would serialise (~1.8 cycles per lane measured).  Such rows
      // are instead accumulated in lane-private registers keyed by the
      // wave-uniform slot `acc.rslot`; the registers are folded into the LDS
      // table only when the slot changes and at the end of the kernel.  Waves
      // whose lanes disagree pay one ballot/readlane/ballot and fall through.
      s << "  bool pend = s >= 0;\n";
      s << "  {\n";
      s << "    const u64 pm = __ballot(pend);\n";
      s << "    if (pm != 0) {\n";
      s << "      const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long) pm) - 1);\n";
      s << "      const int s0 = __builtin_amdgcn_readlane(s, leader);\n";
      s << "      if (__ballot(pend && s == s0) == pm) {\n";
      s << "        if (acc.rslot != s0) { evql_flush_private(lds, acc); acc.rslot = s0; }\n";
      s << "        if (pend) {\n";
      if (kp.need_first_row) s << "          acc.rfirst = row < acc.rfirst ? row : acc.rfirst;\n";
      for (size_t i = 0; i < upd.size(); ++i) {
        const auto& u = upd[i];
        s << "          if (" << u.cond << ") acc.rw[" << i << "] = evql_combine<" << op_name(u.op)
          << ">(acc.rw[" << i << "], " << u.bits << ");\n";
      }
      s << "        }\n";
      s << "        pend = false;\n";
      s << "      }\n    }\n  }\n";
      s << "  if (pend) {\n";
      emit_updates(false, "s", "    ");
      s << "  }\n";
      s << "  const bool spill = live && s < 0;\n";
      s << "  acc.spilled += spill ? 1 : 0;\n";
    } else {
      s << "  const bool spill = live;\n";
    }
    s << "  if (spill) {\n";
    s << "    i64 gs;\n";
    s << "    if (knull) { gs = (i64) A.gcap + 1; A.gtab[(u64) gs * EVQL_WORDS] = 0; }\n";
    s << "    else if (ident == EVQL_EMPTY) { gs = (i64) A.gcap; A.gtab[(u64) gs * EVQL_WORDS] = 0; }\n";
    if (kp.has_ident2()) {
      s << "    else gs = evql_gtab_find2(A.gtab, EVQL_WORDS, A.gcap, ident, ident2, evql_mix64(ident));\n";
    } else {
      s << "    else gs = evql_gtab_find(A.gtab, EVQL_WORDS, A.gcap, ident, evql_mix64(ident));\n";
    }
    s << "    if (gs < 0) { atomicOr(&A.status[0], EVQL_ST_TABLE_FULL); }\n";
    s << "    else {\n";
    emit_updates(true, "gs", "      ");
    s << "    }\n  }\n";
  }
  s << "}\n\n";

  // ---- kernel ------------------------------------------------------------------
  s << "extern \"C\" __global__ void __launch_bounds__(EVQL_BLOCK) evql_scan_agg(const EvqlArgs A) {\n";
  s << "  const u32 tid = threadIdx.x;\n";
  if (grouped && S > 0) {
    s << "  __shared__ u64 lds[EVQL_WORDS * EVQL_LSTRIDE + 1];\n";
    s << "  if (tid == 0) lds[EVQL_WORDS * EVQL_LSTRIDE] = 0;  // rows that found no LDS slot\n";
    s << "  for (u32 i = tid; i < EVQL_LSTRIDE; i += EVQL_BLOCK) {\n";
    s << "    lds[i] = EVQL_EMPTY;\n";
    if (kp.has_ident2()) s << "    lds[1 * EVQL_LSTRIDE + i] = EVQL_EMPTY;\n";
    if (kp.need_first_row) {
      s << "    lds[" << kp.first_row_word() << " * EVQL_LSTRIDE + i] = 0xFFFFFFFFFFFFFFFFull;\n";
    }
    for (int w = 0; w < NW; ++w) {
      s << "    lds[" << (SB + w) << " * EVQL_LSTRIDE + i] = evql_op_identity<"
        << op_name(kp.states[w].op) << ">();\n";
    }
    s << "  }\n  __syncthreads();\n";
  } else {
    s << "  u64* lds = nullptr;\n";
  }
  s << "  EvqlAcc acc;\n  acc.passed = 0;\n  acc.spilled = 0;\n";
  if (grouped && S > 0) {
    s << "  acc.rslot = -1;\n  acc.rfirst = 0xFFFFFFFFFFFFFFFFull;\n";
    for (size_t i = 0; i < updw.size(); ++i) {
      s << "  acc.rw[" << i << "] = evql_op_identity<" << op_name(updw[i].op) << ">();\n";
    }
  }
  if (!grouped) {
    for (int w = 0; w < NW; ++w) {
      s << "  acc.w[" << w << "] = evql_op_identity<" << op_name(kp.states[w].op) << ">();\n";
    }
  }
  s << "  for (u64 t = blockIdx.x; t < A.ntiles; t += gridDim.x) {\n";
  if (grouped) {
    // a full HBM table makes the whole launch void (the host grows it and runs
    // again): stop scanning as soon as any workgroup has reported it
    s << "    if (__hip_atomic_load(&A.status[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & "
         "EVQL_ST_TABLE_FULL) break;\n";
  }
  if (grouped && S > 0) {
    s << "    const bool bypass = *reinterpret_cast<volatile u32*>(&lds[EVQL_WORDS * EVQL_LSTRIDE]) "
         ">= 4u * EVQL_BLOCK;\n";
  } else {
    s << "    const bool bypass = false;\n";
  }
  s << "    const u64 base = (A.tile0 + t) * (u64) EVQL_TILE_ROWS;\n";
  for (int i = 0; i < NC; ++i) {
    s << "    u64 x" << i << "[EVQL_UNROLL][2]; u32 y" << i << "[EVQL_UNROLL][2];\n";
  }
  s << "#pragma unroll\n    for (int u = 0; u < EVQL_UNROLL; ++u) {\n";
  s << "      const u64 r = base + ((u64) (u * EVQL_BLOCK) + tid) * 2;\n";
  for (int i = 0; i < NC; ++i) {
    const ColAccess& c = kp.cols[i];
    switch (c.mode) {
      case ColAccess::PLAIN64:
        s << "      evql_plain64_x2(A.image, A.col[" << i << "].pages, r, x" << i << "[u][0], x" << i
          << "[u][1]);\n";
        break;
      case ColAccess::PLAIN32:
        s << "      evql_plain32_x2(A.image, A.col[" << i << "].pages, r, x" << i << "[u][0], x" << i
          << "[u][1]);\n";
        break;
      case ColAccess::BITPACKED:
        s << "      x" << i << "[u][0] = evql_bitpacked<" << c.bits << ">(A.image, A.col[" << i
          << "].pages, r);\n";
        s << "      x" << i << "[u][1] = evql_bitpacked<" << c.bits << ">(A.image, A.col[" << i
          << "].pages, r + 1);\n";
        break;
      case ColAccess::SOA:
        s << "      evql_soa_x2(A.col[" << i << "].soa, r, x" << i << "[u][0], x" << i << "[u][1]);\n";
        break;
    }
    if (c.has_tags) {
      s << "      { const unsigned short tt = *reinterpret_cast<const unsigned short*>(A.col[" << i
        << "].tags + r); y" << i << "[u][0] = tt & 0xffu; y" << i << "[u][1] = tt >> 8; }\n";
    } else {
      s << "      y" << i << "[u][0] = 0; y" << i << "[u][1] = 0;\n";
    }
  }
  s << "    }\n";
  s << "#pragma unroll\n    for (int u = 0; u < EVQL_UNROLL; ++u) {\n";
  s << "      const u64 r = base + ((u64) (u * EVQL_BLOCK) + tid) * 2;\n";
  for (int j = 0; j < 2; ++j) {
    s << "      evql_row(A, lds, acc, bypass, r + " << j << ", (r + " << j << " >= A.row_begin) && (r + " << j
      << " < A.row_end)";
    for (int i = 0; i < NC; ++i) s << ", x" << i << "[u][" << j << "], y" << i << "[u][" << j << "]";
    s << ");\n";
  }
  s << "    }\n  }\n";

  // ---- epilogue ------------------------------------------------------------------
  s << "  {\n    const u64 p = evql_wave_reduce<EVQL_OP_ADD_U64>(acc.passed);\n";
  s << "    if ((tid & 63u) == 0 && p) atomicAdd(&A.counters[0], p);\n";
  s << "    const u64 sp = evql_wave_reduce<EVQL_OP_ADD_U64>(acc.spilled);\n";
  s << "    if ((tid & 63u) == 0 && sp) atomicAdd(&A.counters[1], sp);\n  }\n";
  if (!grouped) {
    // block reduction of the register accumulators, one atomic per word per block
    s << "  __shared__ u64 red[(EVQL_BLOCK / 64) * " << (NW + 1) << "];\n";
    s << "  const u32 wave = tid >> 6;\n";
    for (int w = 0; w < NW; ++w) {
      s << "  { const u64 v = evql_wave_reduce<" << op_name(kp.states[w].op) << ">(acc.w[" << w
        << "]); if ((tid & 63u) == 0) red[wave * " << (NW + 1) << " + " << w << "] = v; }\n";
    }
    s << "  { const u64 v = evql_wave_reduce<EVQL_OP_ADD_U64>(acc.passed); if ((tid & 63u) == 0) red[wave * "
      << (NW + 1) << " + " << NW << "] = v; }\n";
    s << "  __syncthreads();\n";
    s << "  if (tid == 0) {\n";
    s << "    u64 any = 0;\n";
    s << "    for (u32 k = 0; k < EVQL_BLOCK / 64; ++k) any += red[k * " << (NW + 1) << " + " << NW
      << "];\n";
    s << "    if (any) {\n      A.gtab[0] = 0;\n";
    for (int w = 0; w < NW; ++w) {
      s << "      { u64 v = red[" << w << "]; for (u32 k = 1; k < EVQL_BLOCK / 64; ++k) v = evql_combine<"
        << op_name(kp.states[w].op) << ">(v, red[k * " << (NW + 1) << " + " << w
        << "]); evql_atomic<" << op_name(kp.states[w].op) << ">(&A.gtab[" << (SB + w)
        << "], v); }\n";
    }
    s << "    }\n  }\n";
  } else if (S > 0) {
    s << "  evql_flush_private(lds, acc);\n";
    s << "  __syncthreads();\n";
    s << "  for (u32 s = tid; s < EVQL_LSTRIDE; s += EVQL_BLOCK) {\n";
    s << "    const u64 k = lds[s];\n    if (k == EVQL_EMPTY) continue;\n";
    s << "    i64 gs;\n";
    s << "    if (s == EVQL_LDS_SLOTS) gs = (i64) A.gcap;\n";
    s << "    else if (s == EVQL_LDS_SLOTS + 1) gs = (i64) A.gcap + 1;\n";
    if (kp.has_ident2()) {
      s << "    else gs = evql_gtab_find2(A.gtab, EVQL_WORDS, A.gcap, k, lds[EVQL_LSTRIDE + s], evql_mix64(k));\n";
    } else {
      s << "    else gs = evql_gtab_find(A.gtab, EVQL_WORDS, A.gcap, k, evql_mix64(k));\n";
    }
    s << "    if (gs < 0) { atomicOr(&A.status[0], EVQL_ST_TABLE_FULL); continue; }\n";
    s << "    if (s >= EVQL_LDS_SLOTS) A.gtab[(u64) gs * EVQL_WORDS] = 0;\n";
    if (kp.need_first_row) {
      s << "    evql_atomic<EVQL_OP_MIN_U64>(&A.gtab[(u64) gs * EVQL_WORDS + " << kp.first_row_word()
        << "], lds[" << kp.first_row_word() << " * EVQL_LSTRIDE + s]);\n";
    }
    for (int w = 0; w < NW; ++w) {
      s << "    evql_atomic<" << op_name(kp.states[w].op) << ">(&A.gtab[(u64) gs * EVQL_WORDS + "
        << (SB + w) << "], lds[" << (SB + w) << " * EVQL_LSTRIDE + s]);\n";
    }
    s << "  }\n";
  }
  s << "}\n";
  return s.str();
}

}  // namespace evql
