/*
 * csql_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement, row at a time and in the reference's order, of
 *   VM::evaluate                      sql/runtime/vm.cc:107-157
 *   VM::evaluateVector                sql/runtime/vm.cc:178-229
 *   VM::evaluatePredicateVector       sql/runtime/vm.cc:231-272
 *   stack push/pop helpers            sql/svalue.cc:856-1218
 *   pure functions                    sql/expressions/{boolean,math,conversion}.cc
 *   aggregates count/sum              sql/expressions/aggregate.cc:35-219
 *   FastCSTableScan::nextBatch        sql/CSTableScan.cc:757-995
 *   CSTableScan::fetchNext (NO_AGGREGATION)  sql/CSTableScan.cc:187-541
 *   GroupByExpression::execute/nextBatch     sql/statements/select/groupby.cc:69-220
 *   PartialGroupByExpression::nextBatch      groupby.cc:438-472
 *
 * Build-supplied aggregates (absent from the reference snapshot, SURVEY.md
 * header): sum(float64), min, max, mean.  Their semantics are defined HERE
 * and are what the HIP path is checked against:
 *   - sum_float64 adds the payload of every row in row order regardless of the
 *     tag (exactly what sum_uint64 does, aggregate.cc:184-219); empty => 0.0
 *   - min/max/mean skip STAG_NULL inputs (legacy bodies aggregate.cc:225-441
 *     skip NIL); over zero non-NULL inputs the result is NULL (value 0, tag 1);
 *     mean accumulates (double) value in row order and divides on get;
 *     float min/max also skip NaN inputs (the legacy `first value initialises`
 *     rule would make the result depend on where in the scan a NaN appears)
 * Deliberate non-reproductions of reference *bugs*:
 *   - the evaluateVector X_INPUT shortcut (vm.cc:189-199) is applied only to
 *     programs that are exactly a bare column reference, where it is
 *     equivalent (SURVEY.md header quirk table)
 */
#include "oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define BATCH 1024 /* kOutputBatchSize, CSTableScan.h:46, groupby.h:36 */

/* ---- packed vectors (SVector, svalue.cc:410-517) --------------------------- */
typedef struct {
  int type;
  uint8_t* data;
  size_t size, cap;
} svec_t;

static void sv_reserve(svec_t* v, size_t extra) {
  if (v->size + extra > v->cap) {
    size_t nc = v->cap ? v->cap * 2 : 4096;
    while (nc < v->size + extra) nc *= 2;
    v->data = (uint8_t*) realloc(v->data, nc);
    v->cap = nc;
  }
}
static void sv_append(svec_t* v, const void* p, size_t n) {
  sv_reserve(v, n);
  memcpy(v->data + v->size, p, n);
  v->size += n;
}

/* sql_sizeof, svalue.cc:533-549 */
static size_t elem_size(int type, const void* p) {
  switch (type) {
    case EVQL_T_STRING: {
      uint32_t l;
      memcpy(&l, p, 4);
      return 4 + (size_t) l + 1;
    }
    case EVQL_T_NIL:
      return 1;
    case EVQL_T_BOOL:
      return 2;
    default:
      return 9;
  }
}

/* ---- VM stack (grows downward, vm.cc:77-94) -------------------------------- */
typedef struct {
  uint8_t* data;
  uint8_t* top;
  uint8_t* limit;
} vmstack_t;

static void st_init(vmstack_t* s) {
  size_t n = 1 << 20;
  s->data = (uint8_t*) malloc(n);
  s->limit = s->data + n;
  s->top = s->limit;
}
static void st_push(vmstack_t* s, const void* p, size_t n) {
  if ((size_t) (s->top - s->data) < n) {
    size_t old = s->limit - s->data, used = s->limit - s->top;
    size_t nn = old * 2 + n;
    uint8_t* nd = (uint8_t*) malloc(nn);
    memcpy(nd + nn - used, s->top, used);
    free(s->data);
    s->data = nd;
    s->limit = nd + nn;
    s->top = s->limit - used;
  }
  s->top -= n;
  memcpy(s->top, p, n);
}
static void push_u64(vmstack_t* s, uint64_t v) {
  uint8_t b[9];
  memcpy(b, &v, 8);
  b[8] = 0;
  st_push(s, b, 9);
}
static void push_f64(vmstack_t* s, double v) {
  uint8_t b[9];
  memcpy(b, &v, 8);
  b[8] = 0;
  st_push(s, b, 9);
}
static void push_bool(vmstack_t* s, int v) {
  uint8_t b[2] = {(uint8_t) (v ? 1 : 0), 0};
  st_push(s, b, 2);
}
static void push_nil(vmstack_t* s) {
  uint8_t b = 0;
  st_push(s, &b, 1);
}
static uint64_t pop_u64(vmstack_t* s, uint8_t* tag) {
  uint64_t v;
  memcpy(&v, s->top, 8);
  if (tag) *tag = s->top[8];
  s->top += 9;
  return v;
}
static double pop_f64(vmstack_t* s, uint8_t* tag) {
  double v;
  memcpy(&v, s->top, 8);
  if (tag) *tag = s->top[8];
  s->top += 9;
  return v;
}
static int pop_bool(vmstack_t* s) {
  int v = s->top[0];
  s->top += 2;
  return v;
}
/* strings on the stack: u32 len, bytes, tag */
static void pop_str(vmstack_t* s, const uint8_t** p, uint32_t* len) {
  memcpy(len, s->top, 4);
  *p = s->top + 4;
  s->top += 4 + (size_t) *len + 1;
}

static void push_str(vmstack_t* s, const uint8_t* p, uint32_t len) {
  /* pushString: u32 len, bytes, tag 0 (one element, pushed in one piece) */
  uint8_t* b = (uint8_t*) malloc((size_t) len + 5);
  memcpy(b, &len, 4);
  if (len) memcpy(b + 4, p, len);
  b[4 + len] = 0;
  st_push(s, b, (size_t) len + 5);
  free(b);
}

/* ---- aggregate instances --------------------------------------------------- */
/* count_distinct keeps a std::set<uint64_t> (aggregate.cc:77-80): a hash set
 * here, sorted only when the state is saved */
typedef struct {
  uint64_t* v;
  uint8_t* used;
  size_t n, cap;
} dset_t;

static void dset_insert(dset_t* d, uint64_t x) {
  if ((d->n + 1) * 2 > d->cap) {
    size_t nc = d->cap ? d->cap * 2 : 16;
    uint64_t* nv = (uint64_t*) calloc(nc, sizeof(uint64_t));
    uint8_t* nu = (uint8_t*) calloc(nc, 1);
    for (size_t i = 0; i < d->cap; ++i) {
      if (!d->used[i]) continue;
      size_t p = (size_t) ((d->v[i] * 0x9e3779b97f4a7c15ull) >> 17) & (nc - 1);
      while (nu[p]) p = (p + 1) & (nc - 1);
      nu[p] = 1;
      nv[p] = d->v[i];
    }
    free(d->v);
    free(d->used);
    d->v = nv;
    d->used = nu;
    d->cap = nc;
  }
  size_t p = (size_t) ((x * 0x9e3779b97f4a7c15ull) >> 17) & (d->cap - 1);
  while (d->used[p]) {
    if (d->v[p] == x) return;
    p = (p + 1) & (d->cap - 1);
  }
  d->used[p] = 1;
  d->v[p] = x;
  d->n++;
}

static int cmp_u64(const void* a, const void* b) {
  uint64_t x = *(const uint64_t*) a, y = *(const uint64_t*) b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* the set's values in ascending order (std::set iteration); caller frees */
static uint64_t* dset_sorted(const dset_t* d) {
  uint64_t* out = (uint64_t*) malloc(sizeof(uint64_t) * (d && d->n ? d->n : 1));
  size_t k = 0;
  if (d) {
    for (size_t i = 0; i < d->cap; ++i) {
      if (d->used[i]) out[k++] = d->v[i];
    }
    qsort(out, k, sizeof(uint64_t), cmp_u64);
  }
  return out;
}

typedef struct {
  uint64_t w0; /* count / sum / min / max payload */
  uint64_t w1; /* non-null count for min/max/mean */
  dset_t* ds;  /* count_distinct */
} agg_t;

static void agg_free(agg_t* a) {
  if (a->ds) {
    free(a->ds->v);
    free(a->ds->used);
    free(a->ds);
    a->ds = NULL;
  }
}

static __thread char g_qerr[256];

/* string compare of boolean.cc:150-166: strncmp on the common prefix, then
 * length */
static int str_cmp(const uint8_t* a, uint32_t al, const uint8_t* b, uint32_t bl) {
  uint32_t m = al < bl ? al : bl;
  int c = m ? strncmp((const char*) a, (const char*) b, m) : 0;
  if (c != 0) return c < 0 ? -1 : 1;
  if (al < bl) return -1;
  if (al > bl) return 1;
  return 0;
}

static int call_pure(int64_t fn, vmstack_t* s) {
  int fam = (int) (fn / 16), ts = (int) (fn % 16);
  switch (fam) {
    case EVQL_FAM_LOGICAL_AND: { /* boolean.cc:38: eager */
      int r = pop_bool(s), l = pop_bool(s);
      push_bool(s, l && r);
      return 0;
    }
    case EVQL_FAM_LOGICAL_OR: {
      int r = pop_bool(s), l = pop_bool(s);
      push_bool(s, l || r);
      return 0;
    }
    case EVQL_FAM_NEG: {
      int a = pop_bool(s);
      push_bool(s, !a);
      return 0;
    }
    case EVQL_FAM_CMP:
    case EVQL_FAM_EQ:
    case EVQL_FAM_NEQ:
    case EVQL_FAM_LT:
    case EVQL_FAM_LTE:
    case EVQL_FAM_GT:
    case EVQL_FAM_GTE: {
      int c; /* -1 / 0 / 1; 2 = unordered (NaN) */
      switch (ts) {
        case EVQL_TS_UINT64:
        case EVQL_TS_TIMESTAMP64: {
          uint64_t r = pop_u64(s, NULL), l = pop_u64(s, NULL);
          c = l < r ? -1 : (l > r ? 1 : 0);
          break;
        }
        case EVQL_TS_INT64: {
          int64_t r = (int64_t) pop_u64(s, NULL), l = (int64_t) pop_u64(s, NULL);
          c = l < r ? -1 : (l > r ? 1 : 0);
          break;
        }
        case EVQL_TS_FLOAT64: {
          double r = pop_f64(s, NULL), l = pop_f64(s, NULL);
          c = l < r ? -1 : (l > r ? 1 : (l == r ? 0 : 2));
          break;
        }
        case EVQL_TS_BOOL: {
          int r = pop_bool(s), l = pop_bool(s);
          c = l < r ? -1 : (l > r ? 1 : 0);
          break;
        }
        case EVQL_TS_STRING: {
          const uint8_t *rp, *lp;
          uint32_t rl, ll;
          pop_str(s, &rp, &rl);
          pop_str(s, &lp, &ll);
          c = str_cmp(lp, ll, rp, rl);
          /* eq_string / neq_string compare with memcmp (boolean.cc:237-251,
           * 357-371), which sees past an embedded NUL */
          if (fam == EVQL_FAM_EQ || fam == EVQL_FAM_NEQ) {
            c = (ll == rl && memcmp(lp, rp, ll) == 0) ? 0 : 1;
          }
          break;
        }
        default:
          snprintf(g_qerr, sizeof(g_qerr), "bad compare type slot %d", ts);
          return -1;
      }
      switch (fam) {
        case EVQL_FAM_CMP: /* cmp_float64: else-branch => 0 for NaN */
          push_u64(s, (uint64_t) (int64_t) (c == 2 ? 0 : c));
          break;
        case EVQL_FAM_EQ:
          push_bool(s, c == 0);
          break;
        case EVQL_FAM_NEQ:
          push_bool(s, c != 0);
          break;
        case EVQL_FAM_LT:
          push_bool(s, c == -1);
          break;
        case EVQL_FAM_LTE:
          push_bool(s, c == -1 || c == 0);
          break;
        case EVQL_FAM_GT:
          push_bool(s, c == 1);
          break;
        case EVQL_FAM_GTE:
          push_bool(s, c == 1 || c == 0);
          break;
      }
      return 0;
    }
    case EVQL_FAM_ADD:
    case EVQL_FAM_SUB:
    case EVQL_FAM_MUL:
    case EVQL_FAM_DIV:
    case EVQL_FAM_MOD:
    case EVQL_FAM_POW: {
      if (ts == EVQL_TS_FLOAT64) {
        double r = pop_f64(s, NULL), l = pop_f64(s, NULL), o = 0;
        switch (fam) {
          case EVQL_FAM_ADD: o = l + r; break;
          case EVQL_FAM_SUB: o = l - r; break;
          case EVQL_FAM_MUL: o = l * r; break;
          case EVQL_FAM_DIV: o = l / r; break; /* math.cc:166-170 permitted */
          case EVQL_FAM_MOD: o = fmod(l, r); break;
          case EVQL_FAM_POW: o = pow(l, r); break;
        }
        push_f64(s, o);
        return 0;
      }
      if (ts == EVQL_TS_UINT64) {
        uint64_t r = pop_u64(s, NULL), l = pop_u64(s, NULL), o = 0;
        switch (fam) {
          case EVQL_FAM_ADD: o = l + r; break;
          case EVQL_FAM_SUB: o = l - r; break;
          case EVQL_FAM_MUL: o = l * r; break;
          case EVQL_FAM_DIV:
            if (r == 0) {
              snprintf(g_qerr, sizeof(g_qerr), "division by zero");
              return -1;
            }
            o = l / r;
            break;
          case EVQL_FAM_MOD:
            if (r == 0) {
              snprintf(g_qerr, sizeof(g_qerr), "modulo by zero");
              return -1;
            }
            o = l % r;
            break;
          case EVQL_FAM_POW: /* math.cc:220-224: pow() in double, cast back */
            o = (uint64_t) pow((double) l, (double) r);
            break;
        }
        push_u64(s, o);
        return 0;
      }
      if (ts == EVQL_TS_INT64) {
        int64_t r = (int64_t) pop_u64(s, NULL), l = (int64_t) pop_u64(s, NULL);
        int64_t o = 0;
        switch (fam) {
          case EVQL_FAM_ADD: o = (int64_t) ((uint64_t) l + (uint64_t) r); break;
          case EVQL_FAM_SUB: o = (int64_t) ((uint64_t) l - (uint64_t) r); break;
          case EVQL_FAM_MUL: o = (int64_t) ((uint64_t) l * (uint64_t) r); break;
          case EVQL_FAM_DIV:
            if (r == 0) {
              snprintf(g_qerr, sizeof(g_qerr), "division by zero");
              return -1;
            }
            o = (l == INT64_MIN && r == -1) ? INT64_MIN : l / r;
            break;
          case EVQL_FAM_MOD:
            if (r == 0) {
              snprintf(g_qerr, sizeof(g_qerr), "modulo by zero");
              return -1;
            }
            o = (r == -1) ? 0 : l % r;
            break;
          case EVQL_FAM_POW:
            o = (int64_t) pow((double) l, (double) r);
            break;
        }
        push_u64(s, (uint64_t) o);
        return 0;
      }
      snprintf(g_qerr, sizeof(g_qerr), "bad arithmetic type slot %d", ts);
      return -1;
    }
    case EVQL_FAM_TO_NIL: /* conversion.cc:34-93: pop arg, pushNil (tag 0) */
      switch (ts) {
        case EVQL_TS_BOOL:
          pop_bool(s);
          break;
        case EVQL_TS_STRING: {
          const uint8_t* p;
          uint32_t l;
          pop_str(s, &p, &l);
          break;
        }
        default:
          pop_u64(s, NULL);
      }
      push_nil(s);
      return 0;
    case EVQL_FAM_TO_INT64: /* conversion.cc:96-137 */
      switch (ts) {
        case EVQL_TS_FLOAT64: {
          double v = pop_f64(s, NULL);
          push_u64(s, (uint64_t) (int64_t) v);
          break;
        }
        case EVQL_TS_BOOL: {
          int v = pop_bool(s);
          push_u64(s, (uint64_t) v);
          break;
        }
        default: {
          uint64_t v = pop_u64(s, NULL);
          push_u64(s, v);
        }
      }
      return 0;
    case EVQL_FAM_TO_TIMESTAMP64: /* conversion.cc:223-241 */
      if (ts == EVQL_TS_FLOAT64) {
        double v = pop_f64(s, NULL);
        push_u64(s, (uint64_t) v);
      } else {
        uint64_t v = pop_u64(s, NULL);
        push_u64(s, v);
      }
      return 0;
  }
  /* ---- strings: expressions/string.cc, conversion.cc:140-215 ---------------------------
   * (popped strings are copied first: a push reuses the stack space they sat in) */
  switch (fam) {
    case EVQL_FAM_TO_STRING: { /* sql_tostring, svalue.cc:592-660: NULL tag -> "NULL" */
      char buf[64];
      uint8_t tag = 0;
      switch (ts) {
        case EVQL_TS_FLOAT64: {
          double v = pop_f64(s, &tag);
          /* std::to_string(double) = "%f" */
          char* big = (char*) malloc(512);
          int n = snprintf(big, 512, "%f", v);
          if (tag & EVQL_STAG_NULL) n = snprintf(big, 512, "NULL");
          push_str(s, (const uint8_t*) big, (uint32_t) n);
          free(big);
          return 0;
        }
        case EVQL_TS_BOOL: {
          int v = s->top[0];
          tag = s->top[1];
          s->top += 2;
          snprintf(buf, sizeof(buf), "%s", v ? "true" : "false");
          break;
        }
        case EVQL_TS_STRING: {
          const uint8_t* p;
          uint32_t l;
          pop_str(s, &p, &l); /* (no to_string#string/string; is registered, defaults.cc:110-115) */
          (void) p;
          snprintf(g_qerr, sizeof(g_qerr), "to_string(string) is not registered");
          return -1;
        }
        case EVQL_TS_INT64: {
          int64_t v = (int64_t) pop_u64(s, &tag);
          snprintf(buf, sizeof(buf), "%lld", (long long) v);
          break;
        }
        case EVQL_TS_NIL:
          s->top += 1;
          snprintf(buf, sizeof(buf), "NULL");
          break;
        default: { /* uint64, timestamp64 (to_string_uint64_call for both) */
          uint64_t v = pop_u64(s, &tag);
          snprintf(buf, sizeof(buf), "%llu", (unsigned long long) v);
        }
      }
      if (tag & EVQL_STAG_NULL) snprintf(buf, sizeof(buf), "NULL");
      push_str(s, (const uint8_t*) buf, (uint32_t) strlen(buf));
      return 0;
    }
    case EVQL_FAM_CONCAT:
    case EVQL_FAM_STARTSWITH:
    case EVQL_FAM_ENDSWITH: {
      const uint8_t *rp, *lp;
      uint32_t rl, ll;
      pop_str(s, &rp, &rl);
      pop_str(s, &lp, &ll);
      if (fam == EVQL_FAM_CONCAT) {
        uint8_t* t = (uint8_t*) malloc((size_t) ll + rl + 1);
        memcpy(t, lp, ll);
        memcpy(t + ll, rp, rl);
        push_str(s, t, ll + rl);
        free(t);
      } else if (fam == EVQL_FAM_STARTSWITH) { /* StringUtil::beginsWith */
        push_bool(s, ll >= rl && memcmp(lp, rp, rl) == 0);
      } else {
        push_bool(s, ll >= rl && memcmp(lp + (ll - rl), rp, rl) == 0);
      }
      return 0;
    }
    case EVQL_FAM_LCASE:
    case EVQL_FAM_UCASE:
    case EVQL_FAM_LTRIM:
    case EVQL_FAM_RTRIM: {
      const uint8_t* p;
      uint32_t l;
      pop_str(s, &p, &l);
      uint8_t* t = (uint8_t*) malloc((size_t) l + 1);
      memcpy(t, p, l);
      uint32_t b = 0, e = l;
      if (fam == EVQL_FAM_LCASE || fam == EVQL_FAM_UCASE) {
        for (uint32_t i = 0; i < l; ++i) { /* std::tolower / toupper, "C" locale */
          if (fam == EVQL_FAM_LCASE && t[i] >= 'A' && t[i] <= 'Z') t[i] = (uint8_t) (t[i] - 'A' + 'a');
          if (fam == EVQL_FAM_UCASE && t[i] >= 'a' && t[i] <= 'z') t[i] = (uint8_t) (t[i] - 'a' + 'A');
        }
      } else if (fam == EVQL_FAM_LTRIM) { /* StringUtil::ltrim: ' ' only */
        while (b < e && t[b] == ' ') ++b;
      } else {
        while (e > b && t[e - 1] == ' ') --e;
      }
      push_str(s, t + b, e - b);
      free(t);
      return 0;
    }
    case EVQL_FAM_SUBSTRING: { /* string.cc substring_call */
      int64_t cur = (int64_t) pop_u64(s, NULL);
      const uint8_t* p;
      uint32_t l;
      pop_str(s, &p, &l);
      int64_t len = (int64_t) l;
      if (cur == 0 || len == 0) {
        push_str(s, (const uint8_t*) "", 0);
        return 0;
      }
      if (cur < 0) {
        cur += len;
        if (cur < 0) {
          push_str(s, (const uint8_t*) "", 0);
          return 0;
        }
      } else {
        cur = cur - 1 < len - 1 ? cur - 1 : len - 1;
      }
      uint8_t* t = (uint8_t*) malloc((size_t) (len - cur) + 1);
      memcpy(t, p + cur, (size_t) (len - cur));
      push_str(s, t, (uint32_t) (len - cur));
      free(t);
      return 0;
    }
  }
  snprintf(g_qerr, sizeof(g_qerr), "unknown function id %lld", (long long) fn);
  return -1;
}

static int agg_accumulate(uint32_t fn, agg_t* a, vmstack_t* s) {
  uint8_t tag;
  switch (fn) {
    case EVQL_AGG_COUNT: /* aggregate.cc:35-38: popNil; ++ */
      s->top += 1;
      a->w0 += 1;
      return 0;
    case EVQL_AGG_SUM_UINT64: /* aggregate.cc:184-186 */
    case EVQL_AGG_SUM_INT64:
      a->w0 += pop_u64(s, NULL);
      return 0;
    case EVQL_AGG_SUM_FLOAT64: {
      double v = pop_f64(s, NULL), cur;
      memcpy(&cur, &a->w0, 8);
      cur += v;
      memcpy(&a->w0, &cur, 8);
      return 0;
    }
    case EVQL_AGG_COUNT_DISTINCT_UINT64: { /* aggregate.cc:82-85: insert popUInt64 */
      uint64_t v = pop_u64(s, NULL);
      if (!a->ds) a->ds = (dset_t*) calloc(1, sizeof(dset_t));
      dset_insert(a->ds, v);
      return 0;
    }
    case EVQL_AGG_MIN_UINT64:
    case EVQL_AGG_MAX_UINT64: {
      uint64_t v = pop_u64(s, &tag);
      if (tag & EVQL_STAG_NULL) return 0;
      if (a->w1 == 0 || (fn == EVQL_AGG_MIN_UINT64 ? v < a->w0 : v > a->w0)) {
        a->w0 = v;
      }
      a->w1 += 1;
      return 0;
    }
    case EVQL_AGG_MIN_INT64:
    case EVQL_AGG_MAX_INT64: {
      int64_t v = (int64_t) pop_u64(s, &tag);
      if (tag & EVQL_STAG_NULL) return 0;
      int64_t cur = (int64_t) a->w0;
      if (a->w1 == 0 || (fn == EVQL_AGG_MIN_INT64 ? v < cur : v > cur)) {
        a->w0 = (uint64_t) v;
      }
      a->w1 += 1;
      return 0;
    }
    case EVQL_AGG_MIN_FLOAT64:
    case EVQL_AGG_MAX_FLOAT64: {
      double v = pop_f64(s, &tag), cur;
      if (tag & EVQL_STAG_NULL) return 0;
      /* NaN inputs are skipped like NULLs: with `first value initialises` a NaN
       * would stick or vanish depending on the row order */
      if (v != v) return 0;
      memcpy(&cur, &a->w0, 8);
      if (a->w1 == 0 || (fn == EVQL_AGG_MIN_FLOAT64 ? v < cur : v > cur)) {
        memcpy(&a->w0, &v, 8);
      }
      a->w1 += 1;
      return 0;
    }
    case EVQL_AGG_MEAN_UINT64:
    case EVQL_AGG_MEAN_INT64:
    case EVQL_AGG_MEAN_FLOAT64: {
      double v, cur;
      if (fn == EVQL_AGG_MEAN_FLOAT64) {
        v = pop_f64(s, &tag);
      } else if (fn == EVQL_AGG_MEAN_INT64) {
        v = (double) (int64_t) pop_u64(s, &tag);
      } else {
        v = (double) pop_u64(s, &tag);
      }
      if (tag & EVQL_STAG_NULL) return 0;
      memcpy(&cur, &a->w0, 8);
      cur += v;
      memcpy(&a->w0, &cur, 8);
      a->w1 += 1;
      return 0;
    }
  }
  snprintf(g_qerr, sizeof(g_qerr), "unsupported aggregate %u", fn);
  return -1;
}

static void push_tagged(vmstack_t* s, uint64_t w, uint8_t tag) {
  uint8_t b[9];
  memcpy(b, &w, 8);
  b[8] = tag;
  st_push(s, b, 9);
}

static int agg_get(uint32_t fn, const agg_t* a, vmstack_t* s) {
  switch (fn) {
    case EVQL_AGG_COUNT:
    case EVQL_AGG_SUM_UINT64:
    case EVQL_AGG_SUM_INT64:
    case EVQL_AGG_SUM_FLOAT64:
      push_tagged(s, a->w0, 0);
      return 0;
    case EVQL_AGG_COUNT_DISTINCT_UINT64: /* aggregate.cc:87-90: set size */
      push_tagged(s, a->ds ? (uint64_t) a->ds->n : 0, 0);
      return 0;
    case EVQL_AGG_MIN_UINT64:
    case EVQL_AGG_MAX_UINT64:
    case EVQL_AGG_MIN_INT64:
    case EVQL_AGG_MAX_INT64:
    case EVQL_AGG_MIN_FLOAT64:
    case EVQL_AGG_MAX_FLOAT64:
      if (a->w1 == 0) push_tagged(s, 0, EVQL_STAG_NULL);
      else push_tagged(s, a->w0, 0);
      return 0;
    case EVQL_AGG_MEAN_UINT64:
    case EVQL_AGG_MEAN_INT64:
    case EVQL_AGG_MEAN_FLOAT64:
      if (a->w1 == 0) {
        push_tagged(s, 0, EVQL_STAG_NULL);
      } else {
        double sum, m;
        uint64_t w;
        memcpy(&sum, &a->w0, 8);
        m = sum / (double) a->w1;
        memcpy(&w, &m, 8);
        push_tagged(s, w, 0);
      }
      return 0;
  }
  snprintf(g_qerr, sizeof(g_qerr), "unsupported aggregate %u", fn);
  return -1;
}

/* VM::evaluate, vm.cc:107-157 */
static int vm_evaluate(const evql_program_t* p, uint32_t entry, vmstack_t* s,
                       agg_t* instance, int argc, void** argv) {
  for (uint32_t pc = entry;;) {
    if (pc >= p->code_len) {
      snprintf(g_qerr, sizeof(g_qerr), "pc out of range");
      return -1;
    }
    const evql_instr_t* op = &p->code[pc];
    switch (op->op) {
      case EVQL_X_CALL_PURE:
        if (call_pure(op->arg0, s)) return -1;
        ++pc;
        continue;
      case EVQL_X_CALL_INSTANCE:
        if (op->arg0 == EVQL_INSTANCE_ACCUMULATE) {
          if (agg_accumulate(p->aggregate_fn, instance, s)) return -1;
        } else {
          if (agg_get(p->aggregate_fn, instance, s)) return -1;
        }
        ++pc;
        continue;
      case EVQL_X_LITERAL: {
        /* pushUnboxed, svalue.cc:856-887: NIL pushes nothing */
        if (op->argt != EVQL_T_NIL) {
          const uint8_t* lit = p->static_storage + op->arg0;
          st_push(s, lit, elem_size((int) op->argt, lit));
        }
        ++pc;
        continue;
      }
      case EVQL_X_INPUT:
        if (op->arg0 >= argc) {
          snprintf(g_qerr, sizeof(g_qerr), "invalid input index");
          return -1;
        }
        if (op->argt != EVQL_T_NIL) {
          st_push(s, argv[op->arg0], elem_size((int) op->argt, argv[op->arg0]));
        }
        ++pc;
        continue;
      case EVQL_X_JUMP:
        pc = (uint32_t) op->arg0;
        continue;
      case EVQL_X_CJUMP:
        pc = pop_bool(s) ? (uint32_t) op->arg0 : pc + 1;
        continue;
      case EVQL_X_RETURN:
        return 0;
      default:
        snprintf(g_qerr, sizeof(g_qerr), "bad opcode %u", op->op);
        return -1;
    }
  }
}

/* popVector, svalue.cc:790-821 */
static void pop_vector(vmstack_t* s, svec_t* v) {
  if (v->type == EVQL_T_NIL) return;
  size_t n = elem_size(v->type, s->top);
  sv_append(v, s->top, n);
  s->top += n;
}

/* ---- scan state ------------------------------------------------------------ */
typedef struct {
  orc_table_t* t;
  const evql_plan_desc_t* plan;
  uint32_t ncols;
  orc_column_t** readers;
  svec_t* colbuf; /* column_buffers_ */
  uint64_t remaining, consumed;
  vmstack_t st;
  uint64_t rows_scanned, rows_passed;
  /* nested scan */
  int nested_opened;
} scan_t;

struct orc_result {
  int ncols;
  svec_t* cols;
  uint64_t nrows;
  uint8_t* keys; /* partial mode: 20 B per row */
  size_t keys_cap;
  uint64_t rows_scanned, rows_passed;
};

static int is_bare_input(const evql_program_t* p) {
  return p->code_len >= 2 && p->code[p->method_call].op == EVQL_X_INPUT &&
         p->code[p->method_call + 1].op == EVQL_X_RETURN;
}

/* FastCSTableScan::fetchColumn*, CSTableScan.cc:860-995 */
static int fetch_column(scan_t* sc, uint32_t i, size_t n) {
  svec_t* b = &sc->colbuf[i];
  b->size = 0;
  static __thread uint64_t uv[BATCH];
  static __thread double fv[BATCH];
  static __thread uint8_t pr[BATCH];
  switch (b->type) {
    case EVQL_T_UINT64:
    case EVQL_T_TIMESTAMP64:
      if (orc_column_read_uint(sc->readers[i], n, NULL, NULL, pr, uv)) return -1;
      for (size_t k = 0; k < n; ++k) {
        uint8_t e[9];
        memcpy(e, &uv[k], 8);
        e[8] = pr[k] ? 0 : EVQL_STAG_NULL;
        sv_append(b, e, 9);
      }
      return 0;
    case EVQL_T_FLOAT64:
      if (orc_column_read_float(sc->readers[i], n, NULL, NULL, pr, fv)) return -1;
      for (size_t k = 0; k < n; ++k) {
        uint8_t e[9];
        memcpy(e, &fv[k], 8);
        e[8] = pr[k] ? 0 : EVQL_STAG_NULL;
        sv_append(b, e, 9);
      }
      return 0;
    case EVQL_T_BOOL:
      /* readBoolean: tmp > 0 (column_reader_uint.cc:78-90) */
      if (orc_column_read_uint(sc->readers[i], n, NULL, NULL, pr, uv)) return -1;
      for (size_t k = 0; k < n; ++k) {
        uint8_t e[2];
        e[0] = pr[k] ? (uv[k] > 0) : 0;
        e[1] = pr[k] ? 0 : EVQL_STAG_NULL;
        sv_append(b, e, 2);
      }
      return 0;
    case EVQL_T_STRING: {
      static __thread char* sbuf;
      static __thread uint64_t scap;
      for (size_t k = 0; k < n; ++k) {
        uint64_t r, d, len;
        if (orc_column_read_string_alloc(sc->readers[i], &r, &d, pr, &sbuf,
                                         &scap, &len)) {
          return -1;
        }
        uint32_t l = pr[0] ? (uint32_t) len : 0;
        uint8_t tag = pr[0] ? 0 : EVQL_STAG_NULL;
        sv_append(b, &l, 4);
        sv_append(b, sbuf, l);
        sv_append(b, &tag, 1);
      }
      return 0;
    }
    case EVQL_T_INT64:
      snprintf(g_qerr, sizeof(g_qerr), "illegal column type: INT64");
      return -1;
    default:
      snprintf(g_qerr, sizeof(g_qerr), "illegal column type: NIL");
      return -1;
  }
}

/* FastCSTableScan::nextBatch, CSTableScan.cc:757-858.  out[] = scan select
 * list columns (appended). */
static int flat_next_batch(scan_t* sc, svec_t* out, size_t* nrecords) {
  const evql_plan_desc_t* pl = sc->plan;
  static __thread uint8_t filter_set[BATCH];
  for (;;) {
    if (sc->remaining == 0) {
      *nrecords = 0;
      return 0;
    }
    size_t batch = sc->remaining < BATCH ? (size_t) sc->remaining : BATCH;
    for (uint32_t i = 0; i < sc->ncols; ++i) {
      if (fetch_column(sc, i, batch)) return -1;
    }
    uint64_t batch_offset = sc->consumed;
    sc->remaining -= batch;
    sc->consumed += batch;
    sc->rows_scanned += batch;

    size_t cnt = 0;
    void* cursor[64];
    if (!pl->where) {
      memset(filter_set, 1, batch);
      cnt = batch;
    } else {
      /* evaluatePredicateVector, vm.cc:231-272 */
      for (uint32_t i = 0; i < sc->ncols; ++i) cursor[i] = sc->colbuf[i].data;
      for (size_t n = 0; n < batch; ++n) {
        if (vm_evaluate(pl->where, pl->where->method_call, &sc->st, NULL,
                        (int) sc->ncols, cursor)) {
          return -1;
        }
        uint8_t pred = (uint8_t) pop_bool(&sc->st);
        filter_set[n] = pred;
        cnt += pred; /* note: reference adds the raw byte */
        for (uint32_t i = 0; i < sc->ncols; ++i) {
          cursor[i] = (uint8_t*) cursor[i] +
                      elem_size(sc->colbuf[i].type, cursor[i]);
        }
      }
    }
    if (pl->row_filter_bits) {
      for (size_t i = 0; i < batch; ++i) {
        uint64_t row = batch_offset + i;
        int keep = row < pl->row_filter_len &&
                   ((pl->row_filter_bits[row >> 3] >> (row & 7)) & 1);
        if (!keep && filter_set[i]) {
          filter_set[i] = 0;
          --cnt;
        }
      }
    }
    if (cnt == 0) continue;

    for (uint32_t e = 0; e < pl->n_scan_select; ++e) {
      const evql_program_t* p = &pl->scan_select[e];
      /* evaluateVector, vm.cc:178-229 */
      if (cnt == batch && is_bare_input(p)) {
        const svec_t* src = &sc->colbuf[p->code[p->method_call].arg0];
        sv_append(&out[e], src->data, src->size);
        continue;
      }
      for (uint32_t i = 0; i < sc->ncols; ++i) cursor[i] = sc->colbuf[i].data;
      for (size_t n = 0; n < batch; ++n) {
        if (filter_set[n]) {
          if (vm_evaluate(p, p->method_call, &sc->st, NULL, (int) sc->ncols,
                          cursor)) {
            return -1;
          }
          pop_vector(&sc->st, &out[e]);
        }
        for (uint32_t i = 0; i < sc->ncols; ++i) {
          cursor[i] = (uint8_t*) cursor[i] +
                      elem_size(sc->colbuf[i].type, cursor[i]);
        }
      }
    }
    sc->rows_passed += cnt;
    *nrecords = cnt;
    return 0;
  }
}

/* ---- nested scan: CSTableScan::fetchNext, NO_AGGREGATION ------------------
 * CSTableScan.cc:187-541.  Restated for the strategy the path uses
 * (AggregationStrategy::NO_AGGREGATION): one output row per leaf repetition.
 * Per iteration (one "fetch"):
 *   - every column whose nextRepetitionLevel() >= cur_fetch_level is advanced
 *     by one (r, d, value) slot; its current value becomes NULL when
 *     d < maxDefinitionLevel (:214-330)
 *   - next_level = max over columns of nextRepetitionLevel() (:332-349);
 *     cur_fetch_level = next_level
 *   - WHERE is evaluated on the boxed current values (:351-372); a passing row
 *     is emitted through the select list (:374-399 for NO_AGGREGATION)
 *   - the number of fetches is bounded by the column with the most slots: the
 *     scan ends when every column reader is exhausted (num_records == 0 uses
 *     fetchNextWithoutColumns :543-584)
 * Columns at a shallower repetition depth keep ("repeat") their last value
 * while deeper columns advance. */
typedef struct {
  uint64_t total; /* slots in the column stream */
  uint64_t read;
  uint8_t cur[16]; /* value | tag for fixed types */
  uint8_t* scur;   /* strings */
  size_t scur_len;
  uint32_t rmax;
} ncol_t;

static int nested_scan_all(scan_t* sc, svec_t* out, uint64_t* total_rows);

/* ---- group by --------------------------------------------------------------- */
typedef struct {
  uint8_t key[20];
  agg_t* inst;   /* one per select expr (aggregates) */
  uint8_t** box; /* boxed first-row value per select expr (non-aggregates) */
  size_t* boxlen;
} group_t;

typedef struct {
  group_t* g;
  size_t n, cap;
  int64_t* slots; /* open addressing over key prefix */
  size_t nslots;
} gmap_t;

static uint64_t key_hash(const uint8_t* k) {
  uint64_t h; /* SHA1Hash std::hash: first 8 bytes, util/SHA1.h:104-111 */
  memcpy(&h, k, 8);
  return h;
}

static void gmap_grow(gmap_t* m) {
  size_t ns = m->nslots ? m->nslots * 2 : 1024;
  int64_t* s = (int64_t*) malloc(ns * sizeof(int64_t));
  for (size_t i = 0; i < ns; ++i) s[i] = -1;
  for (size_t i = 0; i < m->n; ++i) {
    size_t p = key_hash(m->g[i].key) & (ns - 1);
    while (s[p] >= 0) p = (p + 1) & (ns - 1);
    s[p] = (int64_t) i;
  }
  free(m->slots);
  m->slots = s;
  m->nslots = ns;
}

static group_t* gmap_get(gmap_t* m, const uint8_t* key, int* is_new) {
  if ((m->n + 1) * 2 > m->nslots) gmap_grow(m);
  size_t p = key_hash(key) & (m->nslots - 1);
  while (m->slots[p] >= 0) {
    group_t* g = &m->g[m->slots[p]];
    if (memcmp(g->key, key, 20) == 0) {
      *is_new = 0;
      return g;
    }
    p = (p + 1) & (m->nslots - 1);
  }
  if (m->n == m->cap) {
    m->cap = m->cap ? m->cap * 2 : 1024;
    m->g = (group_t*) realloc(m->g, m->cap * sizeof(group_t));
  }
  group_t* g = &m->g[m->n];
  memset(g, 0, sizeof(*g));
  memcpy(g->key, key, 20);
  m->slots[p] = (int64_t) m->n;
  m->n++;
  *is_new = 1;
  return g;
}

/* util/io/outputstream.cc appendVarUInt (LEB128) */
static void sv_varuint(svec_t* v, uint64_t x) {
  do {
    uint8_t b = x & 0x7f;
    x >>= 7;
    if (x) b |= 0x80;
    sv_append(v, &b, 1);
  } while (x);
}

/* savestate of each aggregate (aggregate.cc *_save); build-supplied ones:
 * sum_float64 = 8 raw bytes; min/max = varuint count, 8 raw bytes;
 * mean = varuint count, 8 raw bytes (legacy meanExprSave order) */
static void agg_save(uint32_t fn, const agg_t* a, svec_t* v) {
  switch (fn) {
    case EVQL_AGG_COUNT:
    case EVQL_AGG_SUM_UINT64:
    case EVQL_AGG_SUM_INT64:
      sv_varuint(v, a->w0);
      return;
    case EVQL_AGG_SUM_FLOAT64:
      sv_append(v, &a->w0, 8);
      return;
    case EVQL_AGG_COUNT_DISTINCT_UINT64: { /* aggregate.cc:111-117: size, values ascending */
      size_t n = a->ds ? a->ds->n : 0;
      uint64_t* sorted = dset_sorted(a->ds);
      sv_varuint(v, n);
      for (size_t i = 0; i < n; ++i) sv_varuint(v, sorted[i]);
      free(sorted);
      return;
    }
    default:
      sv_varuint(v, a->w1);
      sv_append(v, &a->w0, 8);
  }
}

static void result_reserve_keys(orc_result_t* r, uint64_t rows) {
  if (rows * 20 > r->keys_cap) {
    r->keys_cap = r->keys_cap ? r->keys_cap * 2 : 20 * 1024;
    while (r->keys_cap < rows * 20) r->keys_cap *= 2;
    r->keys = (uint8_t*) realloc(r->keys, r->keys_cap);
  }
}

static void scan_close(scan_t* sc) {
  for (uint32_t i = 0; i < sc->ncols; ++i) {
    if (sc->readers && sc->readers[i]) orc_column_close(sc->readers[i]);
    if (sc->colbuf) free(sc->colbuf[i].data);
  }
  free(sc->readers);
  free(sc->colbuf);
  free(sc->st.data);
}

static int scan_open(scan_t* sc, orc_table_t* t, const evql_plan_desc_t* pl);

/* PartitionCursor::nextBatch, server/sql/partition_cursor.cc:56-81: the current scan's
 * next batch; at its end the next table of the chain is opened */
static int chain_next_batch(scan_t* sc, orc_table_t* const* tables, int ntables,
                            const uint8_t* const* filters, const uint64_t* filter_lens,
                            evql_plan_desc_t* plan_i, int* cur_table, svec_t* out, size_t* n) {
  for (;;) {
    if (flat_next_batch(sc, out, n)) return -1;
    if (*n > 0) return 0;
    if (*cur_table + 1 >= ntables) return 0;
    ++*cur_table;
    if (filters) {
      plan_i->row_filter_bits = filters[*cur_table];
      plan_i->row_filter_len = filters[*cur_table] ? filter_lens[*cur_table] : 0;
    }
    /* (rows_scanned / rows_passed keep counting across the chain) */
    if (scan_open(sc, tables[*cur_table], plan_i)) return -1;
  }
}

/* (re)opens the scan of one table: FastCSTableScan::execute, CSTableScan.cc:726-755 */
static int scan_open(scan_t* sc, orc_table_t* t, const evql_plan_desc_t* pl) {
  sc->t = t;
  sc->plan = pl;
  for (uint32_t i = 0; i < sc->ncols; ++i) {
    if (sc->readers[i]) orc_column_close(sc->readers[i]);
    sc->readers[i] = orc_column_open(t, pl->scan_columns[i]);
    if (!sc->readers[i]) {
      snprintf(g_qerr, sizeof(g_qerr), "column not found: %s", pl->scan_columns[i]);
      return -1;
    }
    sc->colbuf[i].type = (int) pl->scan_column_types[i];
  }
  sc->remaining = orc_table_num_rows(t);
  sc->consumed = 0;
  sc->nested_opened = 0;
  if (pl->row_end && pl->row_end < sc->remaining) sc->remaining = pl->row_end;
  if (pl->row_begin) {
    snprintf(g_qerr, sizeof(g_qerr), "oracle: row_begin unsupported");
    return -1;
  }
  return 0;
}

/* GroupByExpression (or the bare scan) over a CHAIN of tables: the input operator is
 * PartitionCursor (server/sql/partition_cursor.cc:56-81), which hands out the batches
 * of one scan after the other -- newest table first -- each scan with the row filter
 * openNextTable built for it (:197-217, setFilter).  filters[i] == NULL: no filter
 * for table i (needs_filter == false).  One table without a filter override is the
 * plain operator tree. */
static orc_result_t* run_chain(orc_table_t* const* tables, int ntables,
                               const uint8_t* const* filters, const uint64_t* filter_lens,
                               const evql_plan_desc_t* pl0) {
  g_qerr[0] = 0;
  scan_t sc;
  memset(&sc, 0, sizeof(sc));
  evql_plan_desc_t plan_i = *pl0; /* per-table copy: only the row filter differs */
  const evql_plan_desc_t* pl = &plan_i;
  sc.ncols = pl->n_scan_columns;
  if (sc.ncols > 64) {
    snprintf(g_qerr, sizeof(g_qerr), "too many scan columns");
    return NULL;
  }
  sc.readers = (orc_column_t**) calloc(sc.ncols ? sc.ncols : 1, sizeof(void*));
  sc.colbuf = (svec_t*) calloc(sc.ncols ? sc.ncols : 1, sizeof(svec_t));
  st_init(&sc.st);
  int cur_table = 0;
  if (filters) {
    plan_i.row_filter_bits = filters[0];
    plan_i.row_filter_len = filters[0] ? filter_lens[0] : 0;
  }
  if (ntables < 1 || scan_open(&sc, tables[0], pl)) {
    scan_close(&sc);
    return NULL;
  }

  orc_result_t* res = (orc_result_t*) calloc(1, sizeof(orc_result_t));
  int bare_scan = (pl->n_select == 0 && pl->n_group == 0);
  uint32_t nin = pl->n_scan_select;
  svec_t* in = (svec_t*) calloc(nin ? nin : 1, sizeof(svec_t));
  for (uint32_t i = 0; i < nin; ++i) in[i].type = (int) pl->scan_select[i].return_type;
  gmap_t map;
  memset(&map, 0, sizeof(map));
  int failed = 0;

  if (bare_scan) {
    res->ncols = (int) nin;
    res->cols = in;
    if (pl->scan_mode >= EVQL_SCAN_NESTED) {
      if (nested_scan_all(&sc, in, &res->nrows)) failed = 1;
    } else {
      for (;;) {
        size_t n = 0;
        if (chain_next_batch(&sc, tables, ntables, filters, filter_lens, &plan_i, &cur_table, in,
                             &n)) {
          failed = 1;
          break;
        }
        if (n == 0) break;
        res->nrows += n;
      }
    }
  } else {
    /* GroupByExpression::execute, groupby.cc:69-185 */
    int gtypes[64];
    for (uint32_t i = 0; i < pl->n_group; ++i) {
      gtypes[i] = (int) pl->group_exprs[pl->n_group - 1 - i].return_type;
    }
    svec_t* nested_all = NULL;
    uint64_t nested_rows = 0, nested_pos = 0;
    if (pl->scan_mode >= EVQL_SCAN_NESTED) {
      if (nested_scan_all(&sc, in, &nested_rows)) failed = 1;
      nested_all = in;
    }
    void* cursor[64];
    int first_nested = 1;
    while (!failed) {
      size_t n = 0;
      if (pl->scan_mode >= EVQL_SCAN_NESTED) {
        /* the nested scan was materialised in one go; consume it once */
        if (!first_nested) break;
        first_nested = 0;
        n = (size_t) nested_rows;
        (void) nested_pos;
        (void) nested_all;
      } else {
        for (uint32_t i = 0; i < nin; ++i) in[i].size = 0;
        if (chain_next_batch(&sc, tables, ntables, filters, filter_lens, &plan_i, &cur_table, in,
                             &n)) {
          failed = 1;
          break;
        }
      }
      if (n == 0) break;
      for (uint32_t i = 0; i < nin; ++i) cursor[i] = in[i].data;
      for (size_t r = 0; r < n && !failed; ++r) {
        for (uint32_t i = 0; i < pl->n_group; ++i) {
          if (vm_evaluate(&pl->group_exprs[i], pl->group_exprs[i].method_call,
                          &sc.st, NULL, (int) nin, cursor)) {
            failed = 1;
            break;
          }
        }
        if (failed) break;
        /* sql_sizeof_tuple over the reversed types, then SHA1 (:117-135) */
        size_t tlen = 0;
        for (uint32_t i = 0; i < pl->n_group; ++i) {
          tlen += elem_size(gtypes[i], sc.st.top + tlen);
        }
        uint8_t key[20];
        orc_sha1(sc.st.top, tlen, key);
        sc.st.top += tlen;
        int is_new;
        group_t* g = gmap_get(&map, key, &is_new);
        if (is_new) {
          g->inst = (agg_t*) calloc(pl->n_select ? pl->n_select : 1, sizeof(agg_t));
          g->box = (uint8_t**) calloc(pl->n_select ? pl->n_select : 1, sizeof(void*));
          g->boxlen = (size_t*) calloc(pl->n_select ? pl->n_select : 1, sizeof(size_t));
        }
        for (uint32_t e = 0; e < pl->n_select; ++e) {
          const evql_program_t* p = &pl->select_exprs[e];
          if (p->method_accumulate > 0) {
            if (vm_evaluate(p, p->method_accumulate, &sc.st, &g->inst[e],
                            (int) nin, cursor)) {
              failed = 1;
              break;
            }
          } else if (is_new) {
            /* evaluated only on the group's first row (:161-172) */
            if (vm_evaluate(p, p->method_call, &sc.st, NULL, (int) nin, cursor)) {
              failed = 1;
              break;
            }
            if (p->return_type != EVQL_T_NIL) {
              size_t l = elem_size((int) p->return_type, sc.st.top);
              g->box[e] = (uint8_t*) malloc(l);
              memcpy(g->box[e], sc.st.top, l);
              g->boxlen[e] = l;
              sc.st.top += l;
            }
          }
        }
        for (uint32_t i = 0; i < nin; ++i) {
          cursor[i] = (uint8_t*) cursor[i] + elem_size(in[i].type, cursor[i]);
        }
      }
    }
    /* nextBatch, groupby.cc:187-220 (FINAL) / :438-472 (PARTIAL) */
    if (!failed) {
      if (pl->group_mode == EVQL_MODE_PARTIAL) {
        res->ncols = 1;
        res->cols = (svec_t*) calloc(1, sizeof(svec_t));
        res->cols[0].type = EVQL_T_STRING;
        for (size_t gi = 0; gi < map.n; ++gi) {
          group_t* g = &map.g[gi];
          result_reserve_keys(res, gi + 1);
          memcpy(res->keys + 20 * gi, g->key, 20);
          svec_t data;
          memset(&data, 0, sizeof(data));
          for (uint32_t e = 0; e < pl->n_select; ++e) {
            const evql_program_t* p = &pl->select_exprs[e];
            if (p->method_accumulate > 0) {
              agg_save(p->aggregate_fn, &g->inst[e], &data);
            } else {
              /* SValue::encode: u8 type, lenenc(value|tag bytes).  The value sat in an
               * SValue: setData keeps STAG_INLINE in the last byte of the 16-byte inline
               * buffer (svalue.cc:346-368), which is the value's own tag byte when the
               * value is exactly 16 bytes long (a string of 11) */
              uint8_t ty = (uint8_t) p->return_type;
              sv_append(&data, &ty, 1);
              sv_varuint(&data, g->boxlen[e]);
              sv_append(&data, g->box[e], g->boxlen[e]);
              if (g->boxlen[e] == 16) data.data[data.size - 1] |= 0x80;
            }
          }
          uint32_t l = (uint32_t) data.size;
          uint8_t tag = 0;
          sv_append(&res->cols[0], &l, 4);
          sv_append(&res->cols[0], data.data, data.size);
          sv_append(&res->cols[0], &tag, 1);
          free(data.data);
        }
        res->nrows = map.n;
      } else {
        res->ncols = (int) pl->n_select;
        res->cols = (svec_t*) calloc(pl->n_select ? pl->n_select : 1, sizeof(svec_t));
        for (uint32_t e = 0; e < pl->n_select; ++e) {
          res->cols[e].type = (int) pl->select_exprs[e].return_type;
        }
        for (size_t gi = 0; gi < map.n && !failed; ++gi) {
          group_t* g = &map.g[gi];
          for (uint32_t e = 0; e < pl->n_select; ++e) {
            const evql_program_t* p = &pl->select_exprs[e];
            if (p->method_accumulate > 0) {
              if (vm_evaluate(p, p->method_call, &sc.st, &g->inst[e], 0, NULL)) {
                failed = 1;
                break;
              }
              pop_vector(&sc.st, &res->cols[e]);
            } else if (g->box[e]) {
              sv_append(&res->cols[e], g->box[e], g->boxlen[e]);
            }
          }
        }
        res->nrows = map.n;
      }
    }
    for (uint32_t i = 0; i < nin; ++i) free(in[i].data);
    free(in);
  }
  for (size_t gi = 0; gi < map.n; ++gi) {
    for (uint32_t e = 0; e < pl->n_select; ++e) {
      free(map.g[gi].box[e]);
      agg_free(&map.g[gi].inst[e]);
    }
    free(map.g[gi].inst);
    free(map.g[gi].box);
    free(map.g[gi].boxlen);
  }
  free(map.g);
  free(map.slots);
  res->rows_scanned = sc.rows_scanned;
  res->rows_passed = sc.rows_passed;
  scan_close(&sc);
  if (failed) {
    orc_result_free(res);
    return NULL;
  }
  return res;
}

orc_result_t* orc_query_run(orc_table_t* t, const evql_plan_desc_t* pl) {
  return run_chain(&t, 1, NULL, NULL, pl);
}

orc_result_t* orc_query_run_chain(orc_table_t* const* tables, int ntables,
                                  const uint8_t* const* filters, const uint64_t* filter_lens,
                                  const evql_plan_desc_t* pl) {
  if (pl->scan_mode != EVQL_SCAN_FLAT && ntables > 1) {
    snprintf(g_qerr, sizeof(g_qerr), "oracle: chains of nested scans are not restated");
    return NULL;
  }
  return run_chain(tables, ntables, filters, filter_lens, pl);
}

void orc_result_free(orc_result_t* r) {
  if (!r) return;
  for (int i = 0; i < r->ncols; ++i) free(r->cols[i].data);
  free(r->cols);
  free(r->keys);
  free(r);
}
int orc_result_num_columns(const orc_result_t* r) { return r->ncols; }
int orc_result_column_type(const orc_result_t* r, int c) { return r->cols[c].type; }
uint64_t orc_result_num_rows(const orc_result_t* r) { return r->nrows; }
const uint8_t* orc_result_column_data(const orc_result_t* r, int c, size_t* size) {
  *size = r->cols[c].size;
  return r->cols[c].data;
}
const uint8_t* orc_result_group_keys(const orc_result_t* r) { return r->keys; }
uint64_t orc_result_rows_scanned(const orc_result_t* r) { return r->rows_scanned; }
uint64_t orc_result_rows_passed(const orc_result_t* r) { return r->rows_passed; }

const char* orc_query_error(void) { return g_qerr; }

/* nested scan: filled in by csql_nested.inc (kept separate for readability) */
#include "csql_nested.inc"

/* GroupByMergeExpression over partial-aggregate frames (kept separate too) */
#include "csql_merge.inc"

/* OrderByExpression + LimitExpression applied to a result */
#include "csql_order.inc"
