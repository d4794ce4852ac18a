// evql_device.h -- device-side building blocks of the fused
// scan -> filter -> GROUP BY kernels (gfx950 / CDNA4, wave64).
//
// This text is embedded into libevql_mi355x.so and prepended to every
// plan-specific kernel that codegen.cc emits; the result is compiled with
// hiprtc for gfx950.  It is also compiled ahead of time as part of
// aot_kernels.hip (decode / compaction / gather kernels).
//
// Reference semantics implemented here:
//   bit-packed pages      io/cstable/columns/page_reader_bitpacked.cc:30-111 +
//                         libsimdcomp 4-lane vertical layout (simdbitpacking.c)
//   plain pages           page_reader_uint64.cc:50-70, page_reader_uint32.cc,
//                         page_reader_ieee754.cc:38-59
//   group table           replaces std::unordered_map<SHA1Hash, Vector<void*>>
//                         of GroupByExpression (groupby.h:54-62)
//   aggregate states      expressions/aggregate.cc:35-219 (count, sum) and the
//                         build-supplied sum_float64/min/max/mean
#ifndef EVQL_DEVICE_H
#define EVQL_DEVICE_H

typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int u32;
typedef unsigned char u8;

#define EVQL_EMPTY 0xFFFFFFFFFFFFFFFFull
#define EVQL_MAX_COLS 16
#define EVQL_WAVE 64

// status bits written by kernels, read by the host after the launch
#define EVQL_ST_DIV_BY_ZERO 1u
#define EVQL_ST_TABLE_FULL 2u
#define EVQL_ST_MOD_BY_ZERO 4u

// all fields are 8 bytes wide so that host and device agree on the layout
struct EvqlColArg {
  const u64* pages;  // device array: byte offset of each data page in `image`
  const u64* soa;    // pre-decoded values (one u64 per row) or NULL
  const u8* tags;    // pre-decoded tag bytes (STAG_NULL) per row or NULL
  u64 npages;
  // STRING columns compared bytewise: per row (len << 40) | position of the first
  // byte in the column's logical byte stream (pages laid end to end), or NULL
  const u64* strpos;
  // where `pages` offsets count from: the file image, or the private buffer of a
  // column the runtime re-encoded (LEB128 -> narrow bit-packed)
  const u8* base;
};

struct EvqlArgs {
  const u8* image;  // cstable file image in HBM
  u64 row_begin;
  u64 row_end;
  u64 ntiles;      // number of row tiles covering [row_begin,row_end)
  u64 tile0;       // index of the first tile (absolute row / TILE_ROWS)
  const u8* row_filter;  // bit i == 0 drops row i; NULL = none
  u64 row_filter_len;
  u64* gtab;       // HBM group table: word w of slot s at gtab[s*W + w]
  u64 gcap;        // slots (power of two); two extra special slots follow
  u32* status;     // [0] error bits, [1] reserved
  u64* counters;   // [0] rows passed, [1] rows aggregated via LDS overflow
  EvqlColArg col[EVQL_MAX_COLS];
  // count_distinct: one set of (group, value) pairs per aggregate, 3 word planes of
  // pairset_cap[i] (power of two) slots: group identity, value, second identity word
  // / flags
  u64* pairset[4];
  u64 pairset_cap[4];
  // EVQL_FLOAT_SUM_EXACT: per exact sum, 1 / quantum (a power of two) and the bound
  // of |argument|
  double fscale[4];
  double fbound[4];
};

#define EVQL_ST_SUM_RANGE 32u
// the multiple of the quantum nearest to x, split into a signed high part and 31 low
// bits that are added up separately as integers
__device__ __forceinline__ void evql_fix(const EvqlArgs& A, int k, double x, u64& hi, u64& lo) {
  if (!(fabs(x) <= A.fbound[k])) {  // also NaN
    atomicOr(&A.status[0], EVQL_ST_SUM_RANGE);
    x = 0.0;
  }
  const long long q = __double2ll_rn(x * A.fscale[k]);
  hi = (u64) (q >> 31);
  lo = (u64) (q & 0x7fffffffll);
}

// ---------------------------------------------------------------------------
// hashing: slot placement only (identity is the full 64-bit word)
// ---------------------------------------------------------------------------
__device__ __forceinline__ u64 evql_mix64(u64 x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

__device__ __forceinline__ u64 evql_hash_combine(u64 h, u64 v) {
  return evql_mix64(h ^ (v + 0x9e3779b97f4a7c15ULL + (h << 6) + (h >> 2)));
}

// identity of a hashed group key (several keys / strings): two 64-bit words folded over
// the key values in GROUP BY order; value bits `v` (strings: their 64-bit hash), NULL
// tag `g`.  Shared by the generated row function and the conversion of dictionary-coded
// group records into this form (aot_kernels.hip k_dict_records).
#define EVQL_IDENT_SEED1 0x243f6a8885a308d3ull
#define EVQL_IDENT_SEED2 0x13198a2e03707344ull
__device__ __forceinline__ void evql_ident_add(u64& ident, u64& ident2, u64 v, u32 g) {
  ident = evql_hash_combine(ident, v);
  ident = evql_hash_combine(ident, (u64) (g & 1u));
  ident2 = evql_mix64(ident2 * 0x9e3779b97f4a7c15ull + v) ^ (u64) (g & 1u);
}
// (the all-ones pattern marks a free word)
__device__ __forceinline__ u64 evql_ident_word(u64 x) { return x == ~0ull ? ~0ull - 1 : x; }

// ---------------------------------------------------------------------------
// column page access.  Pages are only guaranteed 4-byte aligned in the file
// (a bit-packed page is 4 + 16*b*1024 bytes), so vector loads are declared
// with 4-byte alignment; gfx950 global loads need dword alignment only.
// ---------------------------------------------------------------------------
typedef u32 evql_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef u32 evql_u32x2 __attribute__((ext_vector_type(2), aligned(4)));

// two consecutive 8-byte values (rows r, r+1; r even) of a PLAIN u64/f64 column
__device__ __forceinline__ void evql_plain64_x2(const u8* image, const u64* pages,
                                                u64 r, u64& v0, u64& v1) {
  const u8* p = image + pages[r >> 16] + ((r & 0xffffull) << 3);
  // column pages are streamed once: non-temporal loads (measured on MI355X, 32 GB
  // pure-read stream: 6.0-6.1 TB/s with plain loads, 6.5-6.8 TB/s non-temporal)
  evql_u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const evql_u32x4*>(p));
  v0 = (u64) q.x | ((u64) q.y << 32);
  v1 = (u64) q.z | ((u64) q.w << 32);
}

__device__ __forceinline__ u64 evql_plain64(const u8* image, const u64* pages, u64 r) {
  const u8* p = image + pages[r >> 16] + ((r & 0xffffull) << 3);
  evql_u32x2 q = *reinterpret_cast<const evql_u32x2*>(p);
  return (u64) q.x | ((u64) q.y << 32);
}

// UINT32_PLAIN: 131072 values per 512 KiB page
__device__ __forceinline__ void evql_plain32_x2(const u8* image, const u64* pages,
                                                u64 r, u64& v0, u64& v1) {
  const u8* p = image + pages[r >> 17] + ((r & 0x1ffffull) << 2);
  evql_u32x2 q = __builtin_nontemporal_load(reinterpret_cast<const evql_u32x2*>(p));
  v0 = q.x;
  v1 = q.y;
}

__device__ __forceinline__ u64 evql_plain32(const u8* image, const u64* pages, u64 r) {
  const u8* p = image + pages[r >> 17] + ((r & 0x1ffffull) << 2);
  return *reinterpret_cast<const u32*>(p);
}

// pre-decoded SoA column
__device__ __forceinline__ void evql_soa_x2(const u64* soa, u64 r, u64& v0, u64& v1) {
  typedef u64 evql_u64x2 __attribute__((ext_vector_type(2)));
  const evql_u64x2 q = __builtin_nontemporal_load(reinterpret_cast<const evql_u64x2*>(soa + r));
  v0 = q.x;
  v1 = q.y;
}

// value i of a bit-packed stream of width B (compile-time), libsimdcomp layout:
// block of 128 values = 16*B bytes; value i7: lane l = i7 & 3, k = i7 >> 2,
// bit position p = k*B inside the lane stream; lane word w sits at u32 index
// 4*w + l.  The first page carries a 4-byte max_value header.
// Hides the value range of a decoded bit-packed value from the optimizer.  With
// the range known (<= 24 bits) a following `x % 13` was narrowed to the 24-bit
// float-reciprocal division and came out one too high in the quotient for
// x >= 1.36e7, x = 12 (mod 13) -- remainder "-1" (observed on the 24-bit column of
// tests/test_gpu_parity.py::test_every_bit_width_in_the_fused_kernel).  Plain
// columns never take that path; neither do these after the barrier.
// (not `volatile`: volatile asms are ordered among themselves, which kept the
// unrolled decode blocks from overlapping their loads -- 3.7 ms vs 2.4 ms)
__device__ __forceinline__ u32 evql_opaque(u32 v) {
  asm("" : "+v"(v));
  return v;
}

template <int B>
__device__ __forceinline__ u32 evql_bitpacked(const u8* image, const u64* pages, u64 i) {
  if (B == 0) return 0;
  const u64 page = i >> 17;  // 1024 blocks * 128 values
  const u32 j = (u32) (i & 0x1ffffull);
  const u8* base = image + pages[page] + (page == 0 ? 4 : 0) + (u64) (j >> 7) * (16 * B);
  const u32 i7 = j & 127u, l = i7 & 3u, k = i7 >> 2;
  const u32 p = k * B, w = p >> 5, s = p & 31u;
  const u32* W = reinterpret_cast<const u32*>(base);
  u64 v = (u64) W[4 * w + l] >> s;
  if (s + B > 32) v |= (u64) W[4 * (w + 1) + l] << (32 - s);
  return evql_opaque((u32) (v & (B >= 32 ? 0xffffffffull : ((1ull << (B & 31)) - 1))));
}

// values i (even) and i + 1: adjacent lanes of the same lane word, so one address
// computation, 8-byte loads and 32-bit funnel shifts serve both (the scalar form
// above made a 10-bit key column ALU-bound: 4.9 ms vs 2.6 ms PLAIN per 1e9 rows)
template <int B>
__device__ __forceinline__ void evql_bitpacked_x2(const u8* image, const u64* pages, u64 i,
                                                  u64& a, u64& b) {
  if (B == 0) {
    a = 0;
    b = 0;
    return;
  }
  typedef u32 evql_u32x2 __attribute__((ext_vector_type(2), aligned(4)));
  const u64 page = i >> 17;
  const u32 j = (u32) (i & 0x1ffffull);
  const u8* base = image + pages[page] + (page == 0 ? 4 : 0) + (u64) (j >> 7) * (16 * B);
  const u32 i7 = j & 127u, l = i7 & 3u, k = i7 >> 2;  // l is 0 or 2
  const u32 p = k * B, w = p >> 5, s = p & 31u;
  const u32* W = reinterpret_cast<const u32*>(base) + 4 * w + l;
  const evql_u32x2 lo = *reinterpret_cast<const evql_u32x2*>(W);
  u32 v0, v1;
  if (B == 32) {
    v0 = lo.x;
    v1 = lo.y;
  } else if ((32 % B) == 0) {  // a value never straddles two words
    v0 = lo.x >> s;
    v1 = lo.y >> s;
  } else {
    // the next word of each lane (read unconditionally; when the value does not
    // straddle, its bits are masked away below)
    const evql_u32x2 hi = *reinterpret_cast<const evql_u32x2*>(W + 4);
    v0 = __funnelshift_r(lo.x, hi.x, s);
    v1 = __funnelshift_r(lo.y, hi.y, s);
  }
  const u32 m = B >= 32 ? 0xffffffffu : ((1u << (B & 31)) - 1u);
  a = evql_opaque(v0 & m);
  b = evql_opaque(v1 & m);
}

// runtime-width variant (decode kernels)
__device__ __forceinline__ u32 evql_bitpacked_rt(const u8* image, const u64* pages,
                                                 u32 B, u64 i) {
  if (B == 0) return 0;
  const u64 page = i >> 17;
  const u32 j = (u32) (i & 0x1ffffull);
  const u8* base = image + pages[page] + (page == 0 ? 4 : 0) + (u64) (j >> 7) * (16 * B);
  const u32 i7 = j & 127u, l = i7 & 3u, k = i7 >> 2;
  const u32 p = k * B, w = p >> 5, s = p & 31u;
  const u32* W = reinterpret_cast<const u32*>(base);
  u64 v = (u64) W[4 * w + l] >> s;
  if (s + B > 32) v |= (u64) W[4 * (w + 1) + l] << (32 - s);
  return evql_opaque((u32) (v & (B >= 32 ? 0xffffffffull : ((1ull << B) - 1))));
}

__device__ __forceinline__ bool evql_row_filter(const u8* bits, u64 len, u64 row) {
  return row < len && ((bits[row >> 3] >> (row & 7)) & 1);
}

// ---------------------------------------------------------------------------
// string operands of eq / neq / lt / lte / gt / gte / cmp (boolean.cc:150-166,
// 237-251, 441-452 ...).  A column value lives in the STRING_PLAIN page stream
// (512 KiB pages, bytes may straddle pages); a literal is a constant array.
// ---------------------------------------------------------------------------
struct EvqlStr {
  const u8* base;    // image (column value) or the literal's bytes
  const u64* pages;  // page offsets of the column stream; NULL for a literal
  u64 pos;
  u32 len;
};
__device__ __forceinline__ EvqlStr evql_col_str(const EvqlArgs& A, const EvqlColArg& c, u64 row) {
  const u64 sp = c.strpos[row];
  return EvqlStr{A.image, c.pages, sp & 0xFFFFFFFFFFull, (u32) (sp >> 40)};
}
__device__ __forceinline__ EvqlStr evql_lit_str(const u8* bytes, u32 len) {
  return EvqlStr{bytes, nullptr, 0, len};
}
__device__ __forceinline__ u32 evql_str_byte(const EvqlStr& s, u32 i) {
  const u64 p = s.pos + i;
  return s.pages ? s.base[s.pages[p >> 19] + (p & 0x7ffffu)] : s.base[p];
}
// memcmp-equality (eq_string / neq_string)
__device__ __forceinline__ bool evql_str_eq(const EvqlStr& a, const EvqlStr& b) {
  if (a.len != b.len) return false;
  for (u32 i = 0; i < a.len; ++i) {
    if (evql_str_byte(a, i) != evql_str_byte(b, i)) return false;
  }
  return true;
}
// startswith / endswith (expressions/string.cc:52-74, StringUtil::beginsWith / endsWith:
// std::string::compare over the affix -- bytewise, NULs included)
__device__ __forceinline__ bool evql_str_affix(const EvqlStr& s, const EvqlStr& affix, bool at_end) {
  if (s.len < affix.len) return false;
  const u32 off = at_end ? s.len - affix.len : 0;
  for (u32 i = 0; i < affix.len; ++i) {
    if (evql_str_byte(s, off + i) != evql_str_byte(affix, i)) return false;
  }
  return true;
}
// strncmp over the common prefix (stops at a NUL both sides share), then length
__device__ __forceinline__ int evql_str_cmp(const EvqlStr& a, const EvqlStr& b) {
  const u32 n = a.len < b.len ? a.len : b.len;
  for (u32 i = 0; i < n; ++i) {
    const u32 x = evql_str_byte(a, i), y = evql_str_byte(b, i);
    if (x != y) return x < y ? -1 : 1;
    if (x == 0) break;
  }
  return a.len < b.len ? -1 : (a.len > b.len ? 1 : 0);
}

__device__ __forceinline__ double evql_as_f64(u64 v) { return __longlong_as_double((i64) v); }
__device__ __forceinline__ u64 evql_f64_bits(double v) { return (u64) __double_as_longlong(v); }

// ---------------------------------------------------------------------------
// aggregate state update primitives.  OP codes are shared with the host
// (codegen.cc / runtime.cc):
// ---------------------------------------------------------------------------
#define EVQL_OP_ADD_U64 0
#define EVQL_OP_ADD_F64 1
#define EVQL_OP_MIN_U64 2
#define EVQL_OP_MAX_U64 3
#define EVQL_OP_MIN_I64 4
#define EVQL_OP_MAX_I64 5
#define EVQL_OP_MIN_F64 6
#define EVQL_OP_MAX_F64 7

template <int OP>
__device__ __forceinline__ u64 evql_op_identity() {
  switch (OP) {
    case EVQL_OP_MIN_U64: return 0xFFFFFFFFFFFFFFFFull;
    case EVQL_OP_MAX_U64: return 0ull;
    case EVQL_OP_MIN_I64: return 0x7FFFFFFFFFFFFFFFull;
    case EVQL_OP_MAX_I64: return 0x8000000000000000ull;
    case EVQL_OP_MIN_F64: return 0x7FF0000000000000ull;  // +inf
    case EVQL_OP_MAX_F64: return 0xFFF0000000000000ull;  // -inf
    default: return 0ull;
  }
}

// plain (non-atomic) combine, used for register accumulators and reductions
template <int OP>
__device__ __forceinline__ u64 evql_combine(u64 a, u64 b) {
  switch (OP) {
    case EVQL_OP_ADD_U64: return a + b;
    case EVQL_OP_ADD_F64: return evql_f64_bits(evql_as_f64(a) + evql_as_f64(b));
    case EVQL_OP_MIN_U64: return a < b ? a : b;
    case EVQL_OP_MAX_U64: return a > b ? a : b;
    case EVQL_OP_MIN_I64: return (i64) a < (i64) b ? a : b;
    case EVQL_OP_MAX_I64: return (i64) a > (i64) b ? a : b;
    case EVQL_OP_MIN_F64: return evql_as_f64(b) < evql_as_f64(a) ? b : a;
    case EVQL_OP_MAX_F64: return evql_as_f64(b) > evql_as_f64(a) ? b : a;
  }
  return a;
}

// atomic combine into LDS or global memory (address space is inferred by the
// compiler after inlining: ds_* for LDS, global_atomic_* for HBM)
template <int OP>
__device__ __forceinline__ void evql_atomic(u64* p, u64 v) {
  switch (OP) {
    case EVQL_OP_ADD_U64: atomicAdd(p, v); break;
    case EVQL_OP_ADD_F64: unsafeAtomicAdd(reinterpret_cast<double*>(p), evql_as_f64(v)); break;
    case EVQL_OP_MIN_U64: atomicMin(p, v); break;
    case EVQL_OP_MAX_U64: atomicMax(p, v); break;
    case EVQL_OP_MIN_I64: atomicMin(reinterpret_cast<i64*>(p), (i64) v); break;
    case EVQL_OP_MAX_I64: atomicMax(reinterpret_cast<i64*>(p), (i64) v); break;
    case EVQL_OP_MIN_F64: unsafeAtomicMin(reinterpret_cast<double*>(p), evql_as_f64(v)); break;
    case EVQL_OP_MAX_F64: unsafeAtomicMax(reinterpret_cast<double*>(p), evql_as_f64(v)); break;
  }
}

// ---------------------------------------------------------------------------
// open-addressed group table: word 0 of every slot is the 64-bit identity
// (EVQL_EMPTY = free), claimed with a 64-bit compare-and-swap.
// ---------------------------------------------------------------------------

// A fresh read of an LDS word that other waves update.  NOT a volatile access: address
// space inference leaves volatile loads alone, so `*(volatile u64*) &lds[s]` stayed a
// FLAT load -- it counts in vmcnt as well as lgkmcnt, and the s_waitcnt vmcnt(0) behind
// every probe made each wave wait for ALL of its tile's column loads before the first
// row could be evaluated (ISA of round 2: 16 flat_load_dwordx2 per tile body).  A relaxed
// atomic load is inferred to LDS (ds_read_b64) and the column loads stay in flight while
// rows are evaluated: 10-bit config 2 2.18 -> 1.87 ms, config 3 over 16-bit pages
// 0.386 -> 0.360 ms.
__device__ __forceinline__ u64 evql_lds_peek(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ u32 evql_lds_peek32(const u32* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// LDS table: returns the slot or -1 when no slot was found within MAXP probes
template <int MAXP>
__device__ __forceinline__ int evql_lds_find(u64* keys, u32 mask, u64 ident, u32 h) {
  // Probe 0 is the identity slot (low key bits): dense small keys -- dimension
  // ids, the usual GROUP BY key -- then never collide, and in a wave the probe
  // loop runs as long as its slowest lane.  Everything else continues along the
  // mixed-hash linear chain.
  u32 s = (u32) ident & mask;
  {
    u64 cur = evql_lds_peek(&keys[s]);
    if (cur == ident) return (int) s;
    if (cur == EVQL_EMPTY) {
      u64 old = atomicCAS(&keys[s], EVQL_EMPTY, ident);
      if (old == EVQL_EMPTY || old == ident) return (int) s;
    }
  }
  s = h & mask;
#pragma unroll 1
  for (int probe = 0; probe < MAXP; ++probe) {
    u64 cur = evql_lds_peek(&keys[s]);
    if (cur == ident) return (int) s;
    if (cur == EVQL_EMPTY) {
      u64 old = atomicCAS(&keys[s], EVQL_EMPTY, ident);
      if (old == EVQL_EMPTY || old == ident) return (int) s;
    }
    s = (s + 1) & mask;
  }
  return -1;
}

// global (HBM) table.  Keys never change once written, so a stale plain read
// can only observe EMPTY, which the CAS then corrects.
#define EVQL_GTAB_MAX_PROBE 128
// The HBM table is an array of slots of W consecutive words (identity first):
// all words of a group share one 64-byte line / DRAM page, so the identity probe
// and the state atomics of one row touch one line instead of W word planes.
__device__ __forceinline__ i64 evql_gtab_find(u64* tab, u32 W, u64 cap, u64 ident, u64 h) {
  const u64 mask = cap - 1;
  u64 s = h & mask;
  // the host sizes the table for a load factor <= 1/4, where a chain of 128 is
  // (practically) impossible; running into one means the table is too small:
  // report TABLE_FULL and let the host grow it instead of crawling through it
  const u64 maxp = cap < EVQL_GTAB_MAX_PROBE ? cap : EVQL_GTAB_MAX_PROBE;
#pragma unroll 1
  for (u64 probe = 0; probe < maxp; ++probe) {
    u64* key = tab + s * W;
    u64 cur = __hip_atomic_load(key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == ident) return (i64) s;
    if (cur == EVQL_EMPTY) {
      u64 old = atomicCAS(key, EVQL_EMPTY, ident);
      if (old == EVQL_EMPTY || old == ident) return (i64) s;
    }
    s = (s + 1) & mask;
  }
  return -1;
}

// ---- 128-bit identities (hashed keys): word 0 and word 1 are both claimed with
// a CAS.  Two different keys that agree in the first hash race for the second
// word of the slot; the loser sees a foreign value there and simply moves on
// along the chain, so every (h1, h2) pair ends up owning exactly one slot
// without any lock.
template <int MAXP>
__device__ __forceinline__ int evql_lds_find2(u64* keys, u64* keys2, u32 mask, u64 ident,
                                              u64 ident2, u32 h) {
  u32 s = (u32) ident & mask;
#pragma unroll 1
  for (int probe = 0; probe <= MAXP; ++probe) {
    u64 cur = evql_lds_peek(&keys[s]);
    if (cur == EVQL_EMPTY) cur = atomicCAS(&keys[s], EVQL_EMPTY, ident);
    if (cur == EVQL_EMPTY || cur == ident) {
      u64 c2 = evql_lds_peek(&keys2[s]);
      if (c2 == EVQL_EMPTY) c2 = atomicCAS(&keys2[s], EVQL_EMPTY, ident2);
      if (c2 == EVQL_EMPTY || c2 == ident2) return (int) s;
    }
    s = probe == 0 ? (h & mask) : ((s + 1) & mask);
  }
  return -1;
}

__device__ __forceinline__ i64 evql_gtab_find2(u64* tab, u32 W, u64 cap, u64 ident, u64 ident2,
                                               u64 h) {
  const u64 mask = cap - 1;
  u64 s = h & mask;
  const u64 maxp = cap < EVQL_GTAB_MAX_PROBE ? cap : EVQL_GTAB_MAX_PROBE;
#pragma unroll 1
  for (u64 probe = 0; probe < maxp; ++probe) {
    u64* key = tab + s * W;
    u64 cur = __hip_atomic_load(key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == EVQL_EMPTY) cur = atomicCAS(key, EVQL_EMPTY, ident);
    if (cur == EVQL_EMPTY || cur == ident) {
      u64 c2 = __hip_atomic_load(key + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (c2 == EVQL_EMPTY) c2 = atomicCAS(key + 1, EVQL_EMPTY, ident2);
      if (c2 == EVQL_EMPTY || c2 == ident2) return (i64) s;
    }
    s = (s + 1) & mask;
  }
  return -1;
}

// count_distinct (aggregate.cc:77-137 keeps a std::set per group): one HBM set of
// (group, value) pairs per aggregate; a row adds 1 to its group's state word iff
// it is the one that inserted the pair.  Slot = 3 words, each claimed with its own
// CAS (a slot is accepted only when all three words equal the key; exactly one
// thread sees the last word change from EMPTY).  `flags` = second identity word
// (hashed keys) or NULL-key bit; bits 1 / 2 of it mark an identity / value equal
// to the free-slot marker, which is stored as EMPTY - 1.
#define EVQL_ST_PAIRSET_FULL 8u
__device__ __forceinline__ u64 evql_pairset_insert(const EvqlArgs& A, int which, u64 ident,
                                                   u64 value, u64 flags, bool spare_flag_bits) {
  u64* tab = A.pairset[which];
  const u64 cap = A.pairset_cap[which];
  const u64 mask = cap - 1;
  if (ident == EVQL_EMPTY) {
    ident = EVQL_EMPTY - 1;
    flags = spare_flag_bits ? (flags | 2u) : (flags ^ 0x9e3779b97f4a7c15ull);
  }
  if (value == EVQL_EMPTY) {
    value = EVQL_EMPTY - 1;
    flags = spare_flag_bits ? (flags | 4u) : (flags ^ 0xc2b2ae3d27d4eb4full);
  }
  if (flags == EVQL_EMPTY) flags = EVQL_EMPTY - 1;  // hashed identities only
  u64 s = evql_mix64(evql_hash_combine(evql_hash_combine(ident, value), flags)) & mask;
  const u64 maxp = cap < 256 ? cap : 256;
#pragma unroll 1
  for (u64 probe = 0; probe < maxp; ++probe, s = (s + 1) & mask) {
    u64 c0 = __hip_atomic_load(tab + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c0 == EVQL_EMPTY) c0 = atomicCAS(tab + s, EVQL_EMPTY, ident);
    if (c0 != EVQL_EMPTY && c0 != ident) continue;
    u64 c1 = __hip_atomic_load(tab + cap + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c1 == EVQL_EMPTY) c1 = atomicCAS(tab + cap + s, EVQL_EMPTY, value);
    if (c1 != EVQL_EMPTY && c1 != value) continue;
    u64 c2 = __hip_atomic_load(tab + 2 * cap + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c2 == EVQL_EMPTY) {
      c2 = atomicCAS(tab + 2 * cap + s, EVQL_EMPTY, flags);
      if (c2 == EVQL_EMPTY) return 1;  // this row inserted the pair
    }
    if (c2 == flags) return 0;
  }
  atomicOr(&A.status[0], EVQL_ST_PAIRSET_FULL);
  return 0;
}

// 64-bit wave shuffle
__device__ __forceinline__ u64 evql_shfl_xor(u64 v, int m) {
  u32 lo = (u32) v, hi = (u32) (v >> 32);
  lo = __shfl_xor(lo, m, 64);
  hi = __shfl_xor(hi, m, 64);
  return (u64) lo | ((u64) hi << 32);
}

template <int OP>
__device__ __forceinline__ u64 evql_wave_reduce(u64 v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = evql_combine<OP>(v, evql_shfl_xor(v, m));
  return v;
}

#endif  // EVQL_DEVICE_H
