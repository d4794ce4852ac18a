// aot_kernels.hip -- ahead-of-time gfx950 kernels (see aot_kernels.h).
#include "aot_kernels.h"
#include "evql_device.h"

namespace evql {

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ u64 rt_column_value(const u8* image, const RtColumn& c, u64 i) {
  if (c.base) image = c.base;
  switch (c.mode) {
    case 0: return evql_plain64(image, (const u64*) c.pages, i);
    case 1: return evql_plain32(image, (const u64*) c.pages, i);
    case 2: return evql_bitpacked_rt(image, (const u64*) c.pages, c.bits, i);
    default: return c.soa[i];
  }
}

__device__ __forceinline__ u32 block_excl_scan(u32 v, u32* total);

// byte `pos` of the virtual byte stream over a column's 512 KiB data pages
__device__ __forceinline__ u8 vbyte_fwd(const u8* image, const u64* pages, u64 pos) {
  return image[pages[pos >> 19] + (pos & 0x7ffffull)];
}

// ---- table maintenance -----------------------------------------------------------
__global__ void k_table_init(TableInitArgs a) {
  // slots of `nwords` adjacent words
  const u64 total = a.stride * a.nwords;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (u64) gridDim.x * blockDim.x) {
    a.words[i] = a.identity[i % a.nwords];
  }
}

__global__ void k_table_compact(const u64* words, u64 gcap, u64 stride, u32 nwords,
                                u64* out, u64 max_records, u64* counter) {
  // one atomic per workgroup and round (not per occupied slot: with 1e7 groups a
  // per-slot counter serialised 12 ms of same-address atomics)
  __shared__ u64 base_s;
  const u64 nslots = gcap + 2;
  const u64 rounds = (nslots + (u64) gridDim.x * blockDim.x - 1) / ((u64) gridDim.x * blockDim.x);
  for (u64 it = 0; it < rounds; ++it) {
    const u64 s = (it * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    const u64 k = s < nslots ? words[s * nwords] : EVQL_EMPTY;
    const bool occ = k != EVQL_EMPTY;
    u32 total;
    const u32 ex = block_excl_scan(occ ? 1u : 0u, &total);
    if (threadIdx.x == 0) base_s = total ? atomicAdd(counter, (u64) total) : 0;
    __syncthreads();
    const u64 idx = base_s + ex;
    __syncthreads();
    if (!occ || idx >= max_records) continue;
    u64* rec = out + idx * (nwords + 1);
    rec[0] = s == gcap ? 1ull : (s == gcap + 1 ? 2ull : 0ull);
    rec[1] = s == gcap ? EVQL_EMPTY : k;
    for (u32 w = 1; w < nwords; ++w) rec[1 + w] = words[s * nwords + w];
  }
}

// number of occupied slots (count-only form of the compaction)
__global__ void __launch_bounds__(kBlock) k_table_count(const u64* words, u64 gcap, u32 nwords,
                                                        u64* counter) {
  __shared__ u32 wsum[kBlock / 64];
  const u64 nslots = gcap + 2;
  u32 c = 0;
  for (u64 s = (u64) blockIdx.x * blockDim.x + threadIdx.x; s < nslots;
       s += (u64) gridDim.x * blockDim.x) {
    c += words[s * nwords] != EVQL_EMPTY ? 1u : 0u;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 t = 0;
    for (int i = 0; i < kBlock / 64; ++i) t += wsum[i];
    if (t) atomicAdd((unsigned long long*) counter, (unsigned long long) t);
  }
}

__device__ __forceinline__ void rt_atomic(u32 op, u64* p, u64 v) {
  switch (op) {
    case EVQL_OP_ADD_U64: evql_atomic<EVQL_OP_ADD_U64>(p, v); break;
    case EVQL_OP_ADD_F64: evql_atomic<EVQL_OP_ADD_F64>(p, v); break;
    case EVQL_OP_MIN_U64: evql_atomic<EVQL_OP_MIN_U64>(p, v); break;
    case EVQL_OP_MAX_U64: evql_atomic<EVQL_OP_MAX_U64>(p, v); break;
    case EVQL_OP_MIN_I64: evql_atomic<EVQL_OP_MIN_I64>(p, v); break;
    case EVQL_OP_MAX_I64: evql_atomic<EVQL_OP_MAX_I64>(p, v); break;
    case EVQL_OP_MIN_F64: evql_atomic<EVQL_OP_MIN_F64>(p, v); break;
    case EVQL_OP_MAX_F64: evql_atomic<EVQL_OP_MAX_F64>(p, v); break;
  }
}

// the same combine without the atomic: for slots with one writer per launch (MergeArgs::exclusive)
__device__ __forceinline__ void rt_plain(u32 op, u64* p, u64 v) {
  const u64 o = *p;
  u64 r;
  switch (op) {
    case EVQL_OP_ADD_U64: r = o + v; break;
    case EVQL_OP_ADD_F64: r = evql_f64_bits(evql_as_f64(o) + evql_as_f64(v)); break;
    case EVQL_OP_MIN_U64: r = v < o ? v : o; break;
    case EVQL_OP_MAX_U64: r = v > o ? v : o; break;
    case EVQL_OP_MIN_I64: r = (i64) v < (i64) o ? v : o; break;
    case EVQL_OP_MAX_I64: r = (i64) v > (i64) o ? v : o; break;
    case EVQL_OP_MIN_F64: r = evql_as_f64(v) < evql_as_f64(o) ? v : o; break;
    case EVQL_OP_MAX_F64: r = evql_as_f64(v) > evql_as_f64(o) ? v : o; break;
    default: return;
  }
  *p = r;
}

// slot of a record's group in an HBM table (claimed when new); *fresh = this call claimed it
__device__ __forceinline__ i64 merge_slot(const MergeArgs& a, const u64* rec, bool* fresh) {
  u64* words = (u64*) a.words;
  const u64 kind = rec[0], ident = rec[1];
  *fresh = false;
  if (kind == 1 || kind == 2) {
    // the group of the empty key / the NULL key: at most one record per launch
    const u64 gs = a.gcap + (kind - 1);
    u64* key = words + gs * a.nwords;
    if (a.exclusive) *fresh = __hip_atomic_load(key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == EVQL_EMPTY;
    *key = 0;
    return (i64) gs;
  }
  if (!a.exclusive) {
    return a.has_ident2 ? evql_gtab_find2(words, a.nwords, a.gcap, ident, rec[2], evql_mix64(ident))
                        : evql_gtab_find(words, a.nwords, a.gcap, ident, evql_mix64(ident));
  }
  // (the probe loops of evql_gtab_find / _find2, also telling who claimed the slot)
  const u64 mask = a.gcap - 1;
  u64 s = evql_mix64(ident) & mask;
  const u64 maxp = a.gcap < EVQL_GTAB_MAX_PROBE ? a.gcap : EVQL_GTAB_MAX_PROBE;
#pragma unroll 1
  for (u64 probe = 0; probe < maxp; ++probe) {
    u64* key = words + s * a.nwords;
    u64 cur = __hip_atomic_load(key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool claimed = false;
    if (cur == EVQL_EMPTY) {
      cur = atomicCAS(key, EVQL_EMPTY, ident);
      claimed = cur == EVQL_EMPTY;
    }
    if (claimed || cur == ident) {
      if (!a.has_ident2) {
        *fresh = claimed;
        return (i64) s;
      }
      u64 c2 = __hip_atomic_load(key + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool claimed2 = false;
      if (c2 == EVQL_EMPTY) {
        c2 = atomicCAS(key + 1, EVQL_EMPTY, rec[2]);
        claimed2 = c2 == EVQL_EMPTY;
      }
      if (claimed2 || c2 == rec[2]) {
        *fresh = claimed2;
        return (i64) s;
      }
    }
    s = (s + 1) & mask;
  }
  return -1;
}

// one atomic per wave for the slots its lanes claimed
__device__ __forceinline__ void merge_count_fresh(const MergeArgs& a, bool fresh) {
  if (!a.exclusive || !a.fresh) return;
  const u64 m = __ballot(fresh);
  if (m && (threadIdx.x & 63) == (u32) (__ffsll((long long) m) - 1)) {
    atomicAdd((unsigned long long*) a.fresh, (unsigned long long) __popcll(m));
  }
}

__global__ void k_table_merge(MergeArgs a, const u64* records, u64 n) {
  const u64 step = (u64) gridDim.x * blockDim.x;
  const u64 rounds = (n + step - 1) / step;
  for (u64 it = 0; it < rounds; ++it) {
    const u64 i = it * step + (u64) blockIdx.x * blockDim.x + threadIdx.x;
    bool fresh = false;
    if (i < n) {
      const u64* rec = records + i * (a.nwords + 1);
      const i64 gs = merge_slot(a, rec, &fresh);
      if (gs < 0) {
        atomicOr(&a.status[0], EVQL_ST_TABLE_FULL);
      } else {
        u64* slot = (u64*) a.words + (u64) gs * a.nwords;
        for (u32 w = 1 + a.has_ident2; w < a.nwords; ++w) {
          if (a.exclusive) {
            rt_plain(a.ops[w], &slot[w], rec[1 + w]);
          } else {
            rt_atomic(a.ops[w], &slot[w], rec[1 + w]);
          }
        }
      }
    }
    merge_count_fresh(a, fresh);
  }
}

__global__ void k_gather_rows(const u8* image, const RtColumn* cols, u32 ncols,
                              const u64* rows, u64 n, u64* out_vals, u8* out_tags) {
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    const u64 r = rows[i];
    for (u32 c = 0; c < ncols; ++c) {
      const RtColumn col = cols[c];
      // (an unrecorded first row reads as NULL instead of an address far outside the table)
      out_vals[(u64) c * n + i] = r == EVQL_EMPTY ? 0 : rt_column_value(image, col, r);
      out_tags[(u64) c * n + i] = r == EVQL_EMPTY ? 1 : (col.tags ? col.tags[r] : 0);
    }
  }
}

// ---- exchange of group records between GPUs ---------------------------------------
__device__ __forceinline__ u32 record_owner(const u64* rec, u32 nranks) {
  // the sentinel / NULL key groups (kind != 0) live on rank 0
  return rec[0] != 0 ? 0u : (u32) (evql_mix64(rec[1] ^ 0x2545f4914f6cdd1dull) % nranks);
}

__global__ void __launch_bounds__(kBlock) k_owner_hist(const u64* records, u64 n, u32 rw,
                                                       u32 nranks, u64* counts) {
  __shared__ u32 h[kMaxExchangeRanks];
  if (threadIdx.x < kMaxExchangeRanks) h[threadIdx.x] = 0;
  __syncthreads();
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    atomicAdd(&h[record_owner(records + i * rw, nranks)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < nranks && h[threadIdx.x]) {
    atomicAdd((unsigned long long*) &counts[threadIdx.x], (unsigned long long) h[threadIdx.x]);
  }
}

__global__ void __launch_bounds__(kBlock) k_owner_scatter(const u64* records, u64 n, u32 rw,
                                                          u32 nranks, const u64* starts,
                                                          u64* cursors, u64* out) {
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    const u64* rec = records + i * rw;
    const u32 o = record_owner(rec, nranks);
    const u64 pos = starts[o] + atomicAdd((unsigned long long*) &cursors[o], 1ull);
    u64* dst = out + pos * rw;
    for (u32 w = 0; w < rw; ++w) dst[w] = rec[w];
  }
}

// ---- count_distinct across GPUs: the (group, value, flags) triples of a pair set
// (evql_device.h evql_pairset_insert) travel in their stored form to the rank that owns the
// group and are re-inserted there; the triple that is new to the merged set adds 1 to
// its group's state word (aggregate.cc:119-137 merges the std::sets).
__global__ void __launch_bounds__(kBlock) k_pairset_export(const u64* tab, u64 cap, u64* out,
                                                           u64 max_triples, u64* counter) {
  __shared__ u64 base_s;
  const u64 rounds = (cap + (u64) gridDim.x * blockDim.x - 1) / ((u64) gridDim.x * blockDim.x);
  for (u64 it = 0; it < rounds; ++it) {
    const u64 s = (it * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    // (a triple is complete once its last word is claimed)
    const bool occ = s < cap && tab[2 * cap + s] != EVQL_EMPTY;
    u32 total;
    const u32 ex = block_excl_scan(occ ? 1u : 0u, &total);
    if (threadIdx.x == 0) base_s = total ? atomicAdd((unsigned long long*) counter, (unsigned long long) total) : 0;
    __syncthreads();
    const u64 idx = base_s + ex;
    __syncthreads();
    if (!occ || idx >= max_triples) continue;
    out[idx * 3] = tab[s];
    out[idx * 3 + 1] = tab[cap + s];
    out[idx * 3 + 2] = tab[2 * cap + s];
  }
}

// owner of a triple = owner of its group's record (record_owner)
__device__ __forceinline__ u32 triple_owner(const u64* t, u32 nranks, u32 exact_key) {
  if (exact_key && (t[2] & 3u)) return 0u;  // NULL key / the key 2^64-1: kind != 0
  return (u32) (evql_mix64(t[0] ^ 0x2545f4914f6cdd1dull) % nranks);
}

__global__ void __launch_bounds__(kBlock) k_triple_owner_hist(const u64* triples, u64 n, u32 nranks,
                                                              u32 exact_key, u64* counts) {
  __shared__ u32 h[kMaxExchangeRanks];
  if (threadIdx.x < kMaxExchangeRanks) h[threadIdx.x] = 0;
  __syncthreads();
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    atomicAdd(&h[triple_owner(triples + i * 3, nranks, exact_key)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < nranks && h[threadIdx.x]) {
    atomicAdd((unsigned long long*) &counts[threadIdx.x], (unsigned long long) h[threadIdx.x]);
  }
}

__global__ void __launch_bounds__(kBlock) k_triple_owner_scatter(const u64* triples, u64 n,
                                                                 u32 nranks, u32 exact_key,
                                                                 const u64* starts, u64* cursors,
                                                                 u64* out) {
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    const u64* t = triples + i * 3;
    const u32 o = triple_owner(t, nranks, exact_key);
    const u64 pos = starts[o] + atomicAdd((unsigned long long*) &cursors[o], 1ull);
    out[pos * 3] = t[0];
    out[pos * 3 + 1] = t[1];
    out[pos * 3 + 2] = t[2];
  }
}

// slot of an EXISTING group (every group with a triple had its record merged before)
__device__ __forceinline__ i64 gtab_lookup(const u64* tab, u32 W, u64 cap, u64 ident, u64 ident2,
                                           bool two) {
  const u64 mask = cap - 1;
  u64 s = evql_mix64(ident) & mask;
  const u64 maxp = cap < EVQL_GTAB_MAX_PROBE ? cap : EVQL_GTAB_MAX_PROBE;
  for (u64 probe = 0; probe < maxp; ++probe, s = (s + 1) & mask) {
    const u64 cur = tab[s * W];
    if (cur == EVQL_EMPTY) return -1;
    if (cur == ident && (!two || tab[s * W + 1] == ident2)) return (i64) s;
  }
  return -1;
}

__global__ void __launch_bounds__(kBlock) k_pairset_merge(PairsetMergeArgs a, const u64* triples,
                                                          u64 n) {
  const u64 mask = a.set_cap - 1;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    const u64 ident = triples[i * 3], value = triples[i * 3 + 1], flags = triples[i * 3 + 2];
    // insert the stored form as it is (same claim protocol as evql_pairset_insert)
    u64 s = evql_mix64(evql_hash_combine(evql_hash_combine(ident, value), flags)) & mask;
    const u64 maxp = a.set_cap < 256 ? a.set_cap : 256;
    int fresh = -1;
    for (u64 probe = 0; probe < maxp && fresh < 0; ++probe, s = (s + 1) & mask) {
      u64* set = (u64*) a.set;
      u64 c0 = __hip_atomic_load(set + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (c0 == EVQL_EMPTY) c0 = atomicCAS(set + s, EVQL_EMPTY, ident);
      if (c0 != EVQL_EMPTY && c0 != ident) continue;
      u64 c1 = __hip_atomic_load(set + a.set_cap + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (c1 == EVQL_EMPTY) c1 = atomicCAS(set + a.set_cap + s, EVQL_EMPTY, value);
      if (c1 != EVQL_EMPTY && c1 != value) continue;
      u64 c2 = __hip_atomic_load(set + 2 * a.set_cap + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (c2 == EVQL_EMPTY) {
        c2 = atomicCAS(set + 2 * a.set_cap + s, EVQL_EMPTY, flags);
        if (c2 == EVQL_EMPTY) fresh = 1;
      }
      if (fresh < 0 && c2 == flags) fresh = 0;
    }
    if (fresh < 0) {
      atomicOr(&a.status[0], EVQL_ST_PAIRSET_FULL);
      continue;
    }
    if (!fresh || !a.words) continue;
    // the group this pair belongs to
    i64 gs;
    if (a.key_mode == 1) {  // exact key: flags = NULL-key bit | escape bits
      if (flags & 1u) gs = (i64) a.gcap + 1;
      else if (flags & 2u) gs = (i64) a.gcap;
      else gs = gtab_lookup((const u64*) a.words, a.nwords, a.gcap, ident, 0, false);
    } else if (a.key_mode == 2) {  // hashed key: flags = second identity word, scrambled
      gs = gtab_lookup((const u64*) a.words, a.nwords, a.gcap, ident, flags, true);  // when value was 2^64-1
      if (gs < 0 && value == EVQL_EMPTY - 1) {
        gs = gtab_lookup((const u64*) a.words, a.nwords, a.gcap, ident, flags ^ 0xc2b2ae3d27d4eb4full, true);
      }
    } else {
      gs = gtab_lookup((const u64*) a.words, a.nwords, a.gcap, ident, 0, false);
    }
    if (gs < 0) {
      atomicOr(&a.status[0], EVQL_ST_TABLE_FULL);  // (cannot happen: the record came first)
      continue;
    }
    atomicAdd((unsigned long long*) &a.words[(u64) gs * a.nwords + a.word], 1ull);
  }
}

__global__ void __launch_bounds__(kBlock) k_resolve_records(ResolveArgs a) {
  const u32 ow = a.in_words + a.ncols + 1;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < a.n;
       i += (u64) gridDim.x * blockDim.x) {
    const u64* in = (const u64*) a.in + i * a.in_words;
    u64* out = (u64*) a.out + i * ow;
    for (u32 w = 0; w < a.in_words; ++w) out[w] = in[w];
    const u64 row = in[a.first_row_word];
    out[a.first_row_word] = a.rank_tag | row;
    u64 tags = 0;
    for (u32 c = 0; c < a.ncols; ++c) {
      const RtColumn col = a.cols[c];
      if (row == EVQL_EMPTY) {  // (unrecorded: NULL, not an address far outside the table)
        out[a.in_words + c] = 0;
        tags |= 1ull << c;
        continue;
      }
      out[a.in_words + c] = rt_column_value(a.image, col, row);
      if (col.tags && (col.tags[row] & 1)) tags |= 1ull << c;
    }
    out[a.in_words + a.ncols] = tags;
  }
}

__global__ void __launch_bounds__(kBlock) k_wire_str_sizes(WireStrArgs a) {
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < a.n;
       i += (u64) gridDim.x * blockDim.x) {
    const u64* rec = (const u64*) a.records + i * a.rw;
    const u64 tags = rec[a.tags_word];
    u64 sz = 0;
    for (u32 k = 0; k < a.nstr; ++k) {
      if (!((tags >> a.col[k]) & 1)) sz += rec[a.word[k]] >> 40;
    }
    a.sizes[i] = sz;
  }
}

__global__ void __launch_bounds__(kBlock) k_wire_str_copy(WireStrArgs a) {
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < a.n;
       i += (u64) gridDim.x * blockDim.x) {
    u64* rec = (u64*) a.records + i * a.rw;
    const u64 tags = rec[a.tags_word];
    // this record's owner = the bucket its index falls into
    u32 o = 0;
    while (o + 1 < a.nranks && i >= a.starts[o + 1]) ++o;
    const u64 seg0 = ((const u64*) a.sizes)[a.starts[o]];
    u64 pos = ((const u64*) a.sizes)[i];
    for (u32 k = 0; k < a.nstr; ++k) {
      if ((tags >> a.col[k]) & 1) {
        rec[a.word[k]] = 0;
        continue;
      }
      const u64 sp = rec[a.word[k]];
      const u64 off = sp & kStrOffMask;
      const u32 len = (u32) (sp >> 40);
      for (u32 b = 0; b < len; ++b) a.heap[pos + b] = vbyte_fwd(a.image, (const u64*) a.pages[k], off + b);
      rec[a.word[k]] = ((u64) len << 40) | (pos - seg0);
      pos += len;
    }
  }
}

__global__ void __launch_bounds__(kBlock) k_table_merge_resolved(MergeResolvedArgs a,
                                                                 const u64* records, u64 n) {
  const u32 rw = a.m.nwords + 1;
  const u64 step = (u64) gridDim.x * blockDim.x;
  const u64 rounds = (n + step - 1) / step;
  for (u64 it = 0; it < rounds; ++it) {
    const u64 i = it * step + (u64) blockIdx.x * blockDim.x + threadIdx.x;
    bool fresh = false;
    if (i < n) {
      const u64* rec = records + i * rw;
      const i64 gs = merge_slot(a.m, rec, &fresh);
      if (gs < 0) {
        atomicOr(&a.m.status[0], EVQL_ST_TABLE_FULL);
      } else {
        u64* slot = (u64*) a.m.words + (u64) gs * a.m.nwords;
        bool first = false;
        for (u32 w = 1 + a.m.has_ident2; w < a.state_words; ++w) {
          if (w == a.first_row_word) {
            if (a.m.exclusive) {
              const u64 old = slot[w];
              first = old == EVQL_EMPTY;
              if (rec[1 + w] < old) slot[w] = rec[1 + w];
            } else {
              const u64 old = atomicMin((unsigned long long*) &slot[w], (unsigned long long) rec[1 + w]);
              first = old == EVQL_EMPTY;
            }
          } else if (a.m.exclusive) {
            rt_plain(a.m.ops[w], &slot[w], rec[1 + w]);
          } else {
            rt_atomic(a.m.ops[w], &slot[w], rec[1 + w]);
          }
        }
        if (first) {
          // (one record per group and batch, batches merged one after the other: the
          // first entry of a group has exactly one writer)
          for (u32 c = 0; c <= a.ncols; ++c) {
            u64 v = rec[1 + a.state_words + c];
            if (c < a.ncols && ((a.str_mask >> c) & 1)) {
              v = (v & ~kStrOffMask) | (((v & kStrOffMask) + a.heap_base) & kStrOffMask);
            }
            slot[a.state_words + c] = v;
          }
        }
      }
    }
    merge_count_fresh(a.m, fresh);
  }
}

// ---- bucketed merge (aot_kernels.h BucketScatterArgs / BucketMergeArgs) ------------------
constexpr u64 kBucketSalt = 0x6a09e667f3bcc909ull;
constexpr u32 kBucketTileMax = 2048;

__device__ __forceinline__ u32 bucket_bin(const u64* rec, u32 shift, u32 bins) {
  // (the groups of the empty / the NULL key carry no identity to hash: bucket 0)
  return rec[0] != 0 ? 0u : (u32) (evql_mix64(rec[1] ^ kBucketSalt) >> shift) & (bins - 1);
}

// One pass of the split: every tile of an input region is sorted by bin in the LDS
// (histogram, ranks and the staged records), one cursor add per bin and tile reserves the
// run in the bin's output region, and the runs leave as whole lines.
__global__ void __launch_bounds__(kBlock) k_bucket_scatter(BucketScatterArgs a) {
  extern __shared__ u64 staged[];  // tile x rw words
  __shared__ u32 hist[256], loff[256];
  __shared__ u64 gbase[256];
  __shared__ u8 sbin[kBucketTileMax];
  const u32 tid = threadIdx.x;
  // (a region of the previous pass outgrew its slack: records were dropped, what the regions
  // hold beyond them is stale -- the caller falls back to the table merge, nothing of this
  // attempt is read)
  if (a.in_counts && (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u)) return;
  const u64 vtiles = (u64) a.in_regions * a.tiles_per_region;
  for (u64 vt = blockIdx.x; vt < vtiles; vt += gridDim.x) {
    const u32 reg = (u32) (vt / a.tiles_per_region);
    const u64 t0 = (vt % a.tiles_per_region) * a.tile;
    u64 cnt = a.n_in;
    if (a.in_counts) {
      cnt = a.in_counts[reg];
      if (cnt > a.in_region_cap) cnt = a.in_region_cap;
    }
    if (t0 >= cnt) continue;  // (uniform)
    const u32 nvalid = cnt - t0 < a.tile ? (u32) (cnt - t0) : a.tile;
    const u64* src = (const u64*) a.in + ((u64) reg * a.in_region_cap + t0) * a.rw;
    hist[tid] = 0;
    __syncthreads();
    u32 mybin[kBucketTileMax / kBlock], myrank[kBucketTileMax / kBlock];
#pragma unroll
    for (u32 j = 0; j < kBucketTileMax / kBlock; ++j) {
      const u32 p = j * kBlock + tid;
      mybin[j] = 0;
      myrank[j] = 0;
      if (p < nvalid && p < a.tile) {
        mybin[j] = bucket_bin(src + (u64) p * a.rw, a.shift, a.bins);
        myrank[j] = atomicAdd(&hist[mybin[j]], 1u);
      }
    }
    __syncthreads();
    const u32 c = hist[tid];
    u32 total;
    const u32 ex = block_excl_scan(c, &total);
    loff[tid] = ex;
    gbase[tid] = ~0ull;
    if (c) {
      const u64 outreg = (u64) reg * a.bins + tid;
      const u64 g = atomicAdd((unsigned long long*) &a.out_counts[outreg], (unsigned long long) c);
      if (g + c > a.out_region_cap) {
        atomicOr(a.status, 1u);
      } else {
        gbase[tid] = (outreg * a.out_region_cap + g) * a.rw;
      }
    }
    __syncthreads();
#pragma unroll
    for (u32 j = 0; j < kBucketTileMax / kBlock; ++j) {
      const u32 p = j * kBlock + tid;
      if (p < nvalid && p < a.tile) {
        const u64* rec = src + (u64) p * a.rw;
        const u32 dst = loff[mybin[j]] + myrank[j];
        sbin[dst] = (u8) mybin[j];
        u64* d = staged + (u64) dst * a.rw;
        for (u32 w = 0; w < a.rw; ++w) d[w] = rec[w];
        if (a.nranks && a.str_mask) {
          // string words are (len << 40 | offset into the sender's bytes): the bytes of
          // rank r lie at heap_base[r] of the received heap
          const u64 i = t0 + p;
          u32 r = 0;
          while (r + 1 < a.nranks && i >= a.rank_start[r + 1]) ++r;
          const u64 hb = a.heap_base[r];
          for (u32 col = 0; col < a.ncols; ++col) {
            if (!((a.str_mask >> col) & 1)) continue;
            const u64 v = d[a.first_value_word + col];
            d[a.first_value_word + col] = (v & ~kStrOffMask) | (((v & kStrOffMask) + hb) & kStrOffMask);
          }
        }
      }
    }
    __syncthreads();
    const u32 nwords = nvalid * a.rw;
    for (u32 idx = tid; idx < nwords; idx += kBlock) {
      const u32 p = (u32) (((u64) idx * a.rw_inv) >> 32);
      const u32 w = idx - p * a.rw;
      const u32 b = sbin[p];
      const u64 gb = gbase[b];
      if (gb != ~0ull) a.out[gb + (u64) (p - loff[b]) * a.rw + w] = staged[idx];
    }
    __syncthreads();
  }
}

// slot of a record's group in the LDS table of its bucket; `claim`: insert when new
// (*fresh: this call claimed the slot)
__device__ __forceinline__ int bucket_find(u64* tab, const BucketMergeArgs& a, const u64* rec,
                                           bool claim, bool* fresh) {
  const u64 kind = rec[0];
  *fresh = false;
  if (kind > 2) return -1;  // (not a record)
  if (kind != 0) {
    u64* key = tab + (u64) (a.lds_slots + (u32) kind - 1) * a.mw;
    if (claim) {
      *fresh = atomicExch((unsigned long long*) key, 0ull) == EVQL_EMPTY;
    } else if (evql_lds_peek(key) == EVQL_EMPTY) {
      return -1;
    }
    return (int) (a.lds_slots + (u32) kind - 1);
  }
  const u64 ident = rec[1];
  const u32 mask = a.lds_slots - 1;
  u32 s = (u32) evql_mix64(ident ^ kBucketSalt) & mask;
#pragma unroll 1
  for (u32 probe = 0; probe < a.lds_slots; ++probe) {
    u64* key = tab + (u64) s * a.mw;
    u64 cur = evql_lds_peek(key);
    bool claimed = false;
    if (cur == EVQL_EMPTY) {
      if (!claim) return -1;
      cur = atomicCAS((unsigned long long*) key, (unsigned long long) EVQL_EMPTY, (unsigned long long) ident);
      if (cur == EVQL_EMPTY) {
        cur = ident;
        claimed = true;
      }
    }
    if (cur == ident) {
      if (!a.has_ident2) {
        *fresh = claimed;
        return (int) s;
      }
      // (the protocol of evql_gtab_find2: two keys that agree in the first word race for
      // the second one, the loser moves on along the chain)
      u64 c2 = evql_lds_peek(key + 1);
      bool claimed2 = false;
      if (c2 == EVQL_EMPTY && claim) {
        c2 = atomicCAS((unsigned long long*) (key + 1), (unsigned long long) EVQL_EMPTY,
                       (unsigned long long) rec[2]);
        if (c2 == EVQL_EMPTY) {
          c2 = rec[2];
          claimed2 = true;
        }
      }
      if (c2 == rec[2]) {
        *fresh = claimed2;
        return (int) s;
      }
    }
    s = (s + 1) & mask;
  }
  return -1;
}

// One workgroup per bucket: its records are merged in an LDS table (states with LDS
// atomics, the first row by its smallest (rank << 44 | row) word, whose record then supplies
// the first-row values) and the occupied slots leave as dense records.  The thread that
// claims a slot numbers it, so the bucket reserves its output run with ONE add to the global
// counter (a block scan + an add per 256 slots: 1.6 ms of same-address atomics for 3e4 buckets).
__global__ void __launch_bounds__(kBlock) k_bucket_merge(BucketMergeArgs a) {
  extern __shared__ u64 tab[];  // (lds_slots + 2) x mw words, then one u32 per slot
  __shared__ u64 base_s;
  __shared__ u32 nclaimed;
  const u32 tid = threadIdx.x;
  const u32 rw = a.mw + 1;
  const u32 nslots = a.lds_slots + 2;
  u32* order = (u32*) (tab + (u64) nslots * a.mw);
  if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) return;  // (void attempt)
  for (u32 b = blockIdx.x; b < a.buckets; b += gridDim.x) {
    u64 cnt = a.counts[b];
    if (cnt > a.region_cap) cnt = a.region_cap;
    if (cnt == 0) continue;  // (uniform)
    for (u32 sl = tid; sl < nslots; sl += kBlock) {
      u64* slot = tab + (u64) sl * a.mw;
      for (u32 w = 0; w < a.mw; ++w) slot[w] = a.identity[w];
    }
    if (tid == 0) nclaimed = 0;
    __syncthreads();
    const u64* recs = (const u64*) a.stage + (u64) b * a.region_cap * rw;
    for (u64 i = tid; i < cnt; i += kBlock) {
      const u64* rec = recs + i * rw;
      bool fresh;
      const int s = bucket_find(tab, a, rec, true, &fresh);
      if (s < 0) {
        atomicOr(a.status, 1u);
        continue;
      }
      if (fresh) order[s] = atomicAdd(&nclaimed, 1u);
      u64* slot = tab + (u64) s * a.mw;
      for (u32 w = 1 + a.has_ident2; w < a.state_words; ++w) {
        if (w == a.first_row_word) {
          atomicMin((unsigned long long*) &slot[w], (unsigned long long) rec[1 + w]);
        } else {
          rt_atomic(a.ops[w], &slot[w], rec[1 + w]);
        }
      }
    }
    if (a.first_row_word != 0xffffffffu) {
      __syncthreads();
      for (u64 i = tid; i < cnt; i += kBlock) {
        const u64* rec = recs + i * rw;
        bool fresh;
        const int s = bucket_find(tab, a, rec, false, &fresh);
        if (s < 0) continue;
        u64* slot = tab + (u64) s * a.mw;
        // (rank << 44 | row) is unique: one record per group owns the first row
        if (evql_lds_peek(&slot[a.first_row_word]) != rec[1 + a.first_row_word]) continue;
        for (u32 c = 0; c <= a.ncols; ++c) slot[a.state_words + c] = rec[1 + a.state_words + c];
      }
    }
    __syncthreads();
    if (tid == 0) {
      base_s = nclaimed ? atomicAdd((unsigned long long*) a.out_count, (unsigned long long) nclaimed) : 0;
    }
    __syncthreads();
    const u64 base = base_s;
    for (u32 sl = tid; sl < nslots; sl += kBlock) {
      const u64 k = tab[(u64) sl * a.mw];
      if (k == EVQL_EMPTY) continue;
      const u64 idx = base + order[sl];
      if (idx >= a.out_cap) continue;
      u64* rec = (u64*) a.out + idx * rw;
      rec[0] = sl == a.lds_slots ? 1ull : (sl == a.lds_slots + 1 ? 2ull : 0ull);
      rec[1] = sl == a.lds_slots ? EVQL_EMPTY : k;
      for (u32 w = 1; w < a.mw; ++w) rec[1 + w] = tab[(u64) sl * a.mw + w];
    }
    __syncthreads();
  }
}

// ---- block-wide helpers -----------------------------------------------------------
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    u32 t = __shfl_up(v, d, 64);
    if ((int) (threadIdx.x & 63) >= d) v += t;
  }
  return v;
}

// exclusive scan of one u32 per thread across a 256-thread block; *total out
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32* total) {
  __shared__ u32 wsum[kBlock / 64];
  const u32 incl = wave_incl_scan(v);
  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  u32 off = 0, tot = 0;
  for (u32 w = 0; w < kBlock / 64; ++w) {
    if (w < wave) off += wsum[w];
    tot += wsum[w];
  }
  *total = tot;
  return off + incl - v;
}

// ---- definition levels -> tags ---------------------------------------------------
// tile = 2048 rows, 8 consecutive rows per thread
__global__ void __launch_bounds__(kBlock) k_dlevel_tags(const u8* image, const u64* pages,
                                                        u32 dbits, u32 dmax, u64 nrows,
                                                        u8* tags, u64* tile_counts) {
  const u64 tile = blockIdx.x;
  const u64 r0 = tile * kDecodeTile + (u64) threadIdx.x * 8;
  u32 cnt = 0;
  u64 packed = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const u64 r = r0 + j;
    u32 tag = 1;
    if (r < nrows) {
      const u32 d = evql_bitpacked_rt(image, pages, dbits, r);
      tag = d == dmax ? 0 : 1;
      cnt += 1 - tag;
    }
    packed |= (u64) tag << (8 * j);
  }
  // tags buffer is padded to a tile multiple
  *reinterpret_cast<u64*>(tags + r0) = packed;
  u32 total;
  block_excl_scan(cnt, &total);
  if (threadIdx.x == 0) tile_counts[tile] = total;
}

__global__ void __launch_bounds__(1024) k_exclusive_scan(u64* data, u64 n, u64* total) {
  __shared__ u64 wsum[16];
  __shared__ u64 carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (u64 base = 0; base < n; base += 1024) {
    const u64 i = base + threadIdx.x;
    const u64 v = i < n ? data[i] : 0;
    u64 incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      u32 lo = (u32) incl, hi = (u32) (incl >> 32);
      lo = __shfl_up(lo, d, 64);
      hi = __shfl_up(hi, d, 64);
      if ((int) (threadIdx.x & 63) >= d) incl += (u64) lo | ((u64) hi << 32);
    }
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u64 off = carry_s, tot = 0;
    for (u32 w = 0; w < 16; ++w) {
      if (w < wave) off += wsum[w];
      tot += wsum[w];
    }
    if (i < n) data[i] = off + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) carry_s += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total) *total = carry_s;
}

__global__ void __launch_bounds__(kBlock) k_expand_nullable(const u8* image, RtColumn src,
                                                            const u8* tags,
                                                            const u64* tile_offsets, u64 nrows,
                                                            u64* values) {
  const u64 tile = blockIdx.x;
  const u64 r0 = tile * kDecodeTile + (u64) threadIdx.x * 8;
  const u64 packed = *reinterpret_cast<const u64*>(tags + r0);
  u32 cnt = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) cnt += ((packed >> (8 * j)) & 1) ? 0 : (r0 + j < nrows ? 1 : 0);
  u32 total;
  u32 li = block_excl_scan(cnt, &total);
  // row -> index of its value inside the tile (0xffff = undefined), staged in LDS so
  // that the gather and the stores below run with consecutive lanes on consecutive
  // rows (8 rows per lane made both strided: 0.57 ms per 4.3e7 slots)
  __shared__ unsigned short vidx[kDecodeTile];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool defined = !((packed >> (8 * j)) & 1) && r0 + j < nrows;
    vidx[threadIdx.x * 8 + j] = defined ? (unsigned short) li++ : (unsigned short) 0xffff;
  }
  __syncthreads();
  const u64 base = tile_offsets[tile];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const u32 row = (u32) k * kBlock + threadIdx.x;
    const u32 x = vidx[row];
    values[tile * kDecodeTile + row] = x == 0xffffu ? 0 : rt_column_value(image, src, base + x);
  }
}

// ---- LEB128 -------------------------------------------------------------------------
__device__ __forceinline__ u8 vbyte(const u8* image, const u64* pages, u64 pos) {
  return image[pages[pos >> 19] + (pos & 0x7ffffull)];
}

// chunk = 4096 bytes, 16 bytes per thread
__global__ void __launch_bounds__(kBlock) k_leb128_count(const u8* image, const u64* pages,
                                                         u64 nbytes, u64* chunk_counts) {
  const u64 p0 = (u64) blockIdx.x * kLebChunk + (u64) threadIdx.x * 16;
  u32 cnt = 0;
  if (p0 < nbytes) {
    const u8* p = image + pages[p0 >> 19] + (p0 & 0x7ffffull);
    const evql_u32x4 q = *reinterpret_cast<const evql_u32x4*>(p);
    const u32 w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) cnt += 4 - __popc(w[k] & 0x80808080u);
  }
  u32 total;
  block_excl_scan(cnt, &total);
  if (threadIdx.x == 0) chunk_counts[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kBlock) k_leb128_decode(const u8* image, const u64* pages,
                                                          u64 nbytes, const u64* chunk_offsets,
                                                          u64 nvalues, u64* values) {
  const u64 p0 = (u64) blockIdx.x * kLebChunk + (u64) threadIdx.x * 16;
  // bytes p0 .. p0+31 as four little-endian words: the thread's own 16 bytes and
  // the 16 behind them (a value starting in the first may run 9 bytes into the
  // second).  Everything below indexes them statically: a byte array indexed by a
  // loop variable went to scratch memory and the pass ran at 0.64 TB/s.
  u64 w0 = 0, w1 = 0, w2 = 0, w3 = 0;
  u32 cnt = 0;
  if (p0 < nbytes) {
    const u8* p = image + pages[p0 >> 19] + (p0 & 0x7ffffull);
    const evql_u32x4 q = *reinterpret_cast<const evql_u32x4*>(p);
    w0 = (u64) q.x | ((u64) q.y << 32);
    w1 = (u64) q.z | ((u64) q.w << 32);
    const u64 p1 = p0 + 16;
    if (p1 < nbytes) {
      const u8* pn = image + pages[p1 >> 19] + (p1 & 0x7ffffull);
      const evql_u32x4 qn = *reinterpret_cast<const evql_u32x4*>(pn);
      w2 = (u64) qn.x | ((u64) qn.y << 32);
      w3 = (u64) qn.z | ((u64) qn.w << 32);
    }
    cnt = (u32) __popcll(~w0 & 0x8080808080808080ull) + (u32) __popcll(~w1 & 0x8080808080808080ull);
  }
  u32 total;
  u64 idx = chunk_offsets[blockIdx.x] + block_excl_scan(cnt, &total);
  if (p0 >= nbytes) return;
  bool prev_term = p0 == 0 ? true : !(vbyte(image, pages, p0 - 1) & 0x80);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    // 16-byte window starting at byte k: x0 = bytes k..k+7, x1 = bytes k+8..k+15
    const int s = 8 * (k & 7);
    const u64 a = k < 8 ? w0 : w1, b = k < 8 ? w1 : w2, c = k < 8 ? w2 : w3;
    const u64 x0 = s ? (a >> s) | (b << (64 - s)) : a;
    const u64 x1 = s ? (b >> s) | (c << (64 - s)) : b;
    if (prev_term && idx < nvalues) {
      // gather the 7-bit groups of 8 bytes into 56 bits
      u64 g = x0 & 0x7f7f7f7f7f7f7f7full;
      g = (g & 0x00ff00ff00ff00ffull) | ((g & 0xff00ff00ff00ff00ull) >> 1);
      g = (g & 0x0000ffff0000ffffull) | ((g & 0xffff0000ffff0000ull) >> 2);
      g = (g & 0x00000000ffffffffull) | ((g & 0xffffffff00000000ull) >> 4);
      const u64 stop = ~x0 & 0x8080808080808080ull;  // bytes without continuation bit
      u64 v;
      if (stop) {
        const int n = (__ffsll((long long) stop) >> 3);  // bytes of this value, 1..8
        v = g & ((1ull << (7 * n)) - 1);
      } else {  // 9 or 10 bytes
        v = g | ((x1 & 0x7f) << 56);
        if (x1 & 0x80) v |= ((x1 >> 8) & 0x7f) << 63;
      }
      values[idx] = v;
    }
    prev_term = !((x0 >> 7) & 1);
    if (prev_term) ++idx;
  }
}

__global__ void k_string_hash(const u8* image, const u64* pages, const u64* strpos, u64 n,
                              u64* out) {
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    const u64 sp = strpos[i];
    const u64 off = sp & 0xffffffffffull;
    const u32 len = (u32) (sp >> 40);
    u64 h = 0xcbf29ce484222325ull ^ ((u64) len * 0x9e3779b97f4a7c15ull);
    for (u32 k = 0; k < len; ++k) {
      h ^= vbyte(image, pages, off + k);
      h *= 0x100000001b3ull;
    }
    out[i] = evql_mix64(h);
  }
}

__global__ void __launch_bounds__(kBlock) k_column_abs_max(const u8* image, RtColumn col, u64 n,
                                                           u32 is_float, u64* out) {
  u64 m = 0;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    u64 v = rt_column_value(image, col, i);
    if (is_float) v &= 0x7fffffffffffffffull;  // fabs
    m = v > m ? v : m;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    u32 lo = __shfl_xor((u32) m, d, 64), hi = __shfl_xor((u32) (m >> 32), d, 64);
    const u64 o = (u64) lo | ((u64) hi << 32);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax((unsigned long long*) out, (unsigned long long) m);
}

__global__ void __launch_bounds__(kBlock) k_max_u64(const u64* values, u64 n, u64* out) {
  u64 m = 0;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    const u64 v = values[i];
    m = v > m ? v : m;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    u32 lo = __shfl_xor((u32) m, d, 64), hi = __shfl_xor((u32) (m >> 32), d, 64);
    const u64 o = (u64) lo | ((u64) hi << 32);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax((unsigned long long*) out, (unsigned long long) m);
}

// ---- STRING_PLAIN value boundaries (see aot_kernels.h) --------------------------------
typedef unsigned short u16;

// stages chunk c (+16 look-ahead bytes, zeros behind the stream) in LDS
__device__ __forceinline__ void str_stage_chunk(const StrScanArgs& a, u64 c, u8* bytes, u32 nthreads) {
  const u64 base = c * kStrChunk;
  for (u32 t = threadIdx.x; t < kStrChunk / 16 + 1; t += nthreads) {
    const u64 p0 = base + (u64) t * 16;
    evql_u32x4 q = {0, 0, 0, 0};
    if (p0 < a.nbytes) {
      q = *reinterpret_cast<const evql_u32x4*>(a.image + a.pages[p0 >> 19] + (p0 & 0x7ffffull));
    }
    *reinterpret_cast<evql_u32x4*>(bytes + t * 16) = q;
  }
}

// next(p) of a value assumed to start at chunk offset p, clamped to 0xffff (= too far
// for a table entry).  Three header bytes decide: a longer header means >= 2^21 bytes
__device__ __forceinline__ u32 str_next(const u8* bytes, u32 p) {
  const u32 b0 = bytes[p];
  u32 len = b0 & 0x7f, hdr = 1;
  if (b0 & 0x80) {
    const u32 b1 = bytes[p + 1];
    len |= (b1 & 0x7f) << 7;
    hdr = 2;
    if (b1 & 0x80) {
      const u32 b2 = bytes[p + 2];
      len |= (b2 & 0x7f) << 14;
      hdr = 3;
      if (b2 & 0x80) len = 0xffffff;
    }
  }
  const u32 nxt = p + hdr + len;
  return nxt < 0xffffu ? nxt : 0xffffu;
}

__global__ void __launch_bounds__(kBlock) k_str_chunk_tables(StrScanArgs a) {
  __shared__ __attribute__((aligned(16))) u8 bytes[kStrChunk + 16];
  __shared__ u16 J[2][kStrChunk];
  __shared__ u16 H[2][kStrChunk];
  for (u64 c = blockIdx.x; c < a.nchunks; c += gridDim.x) {
    str_stage_chunk(a, c, bytes, kBlock);
    __syncthreads();
#pragma unroll 4
    for (u32 k = 0; k < kStrChunk / kBlock; ++k) {
      const u32 p = threadIdx.x + k * kBlock;
      J[0][p] = (u16) str_next(bytes, p);
      H[0][p] = 1;
    }
    __syncthreads();
    int cur = 0;
    // pointer doubling: after round r, J = position after 2^r values or the first
    // position at/behind the chunk's end; H = values started inside the chunk
    for (int round = 0; round < 12; ++round) {
      int any = 0;
#pragma unroll 4
      for (u32 k = 0; k < kStrChunk / kBlock; ++k) {
        const u32 p = threadIdx.x + k * kBlock;
        u32 j = J[cur][p], h = H[cur][p];
        if (j < kStrChunk) {
          h += H[cur][j];
          j = J[cur][j];
          any |= j < kStrChunk;
        }
        J[cur ^ 1][p] = (u16) j;
        H[cur ^ 1][p] = (u16) h;
      }
      cur ^= 1;
      if (!__syncthreads_or(any)) break;
    }
    for (u32 e = threadIdx.x; e < kStrEntries; e += kBlock) {
      const u32 j = J[cur][e];
      a.exits[c * kStrEntries + e] = j == 0xffffu ? (u16) kStrNone : (u16) (j - kStrChunk);
      a.hops[c * kStrEntries + e] = H[cur][e];
    }
    __syncthreads();
  }
}

// group g = kStrGroup consecutive chunks: the chunk tables composed for every entry
__global__ void __launch_bounds__(kStrEntries) k_str_group_compose(StrScanArgs a) {
  const u64 g = blockIdx.x;
  const u64 c0 = g * kStrGroup;
  const u64 nc = a.nchunks - c0 < kStrGroup ? a.nchunks - c0 : kStrGroup;
  u64 pos = threadIdx.x;  // relative to the group's first byte
  u32 cnt = 0;
  bool over = false;
  for (u64 j = 0; j < nc; ++j) {
    const u64 cb = j * kStrChunk;
    if (pos >= cb + kStrChunk) continue;  // a long value covers this chunk entirely
    const u64 rel = pos - cb;
    if (rel >= kStrEntries) {
      over = true;
      break;
    }
    const u32 x = a.exits[(c0 + j) * kStrEntries + rel];
    if (x == kStrNone) {
      over = true;
      break;
    }
    cnt += a.hops[(c0 + j) * kStrEntries + rel];
    pos = cb + kStrChunk + x;
  }
  a.gexit[g * kStrEntries + threadIdx.x] = over ? 0xffffffffu : (u32) (pos - nc * kStrChunk);
  a.ghops[g * kStrEntries + threadIdx.x] = cnt;
}

// the true entry of every group: one short dependent chain
__global__ void k_str_chain(StrScanArgs a) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const u64 ngroups = (a.nchunks + kStrGroup - 1) / kStrGroup;
  u64 pos = 0, vals = 0;
  for (u64 g = 0; g < ngroups; ++g) {
    const u64 gb = g * kStrGroup * (u64) kStrChunk;
    const u64 nc = a.nchunks - g * kStrGroup < kStrGroup ? a.nchunks - g * kStrGroup : kStrGroup;
    const u64 ge = gb + nc * kStrChunk;
    if (vals >= a.nvalues) {  // everything behind the last value is padding
      a.gentry[g] = ~0ull;
      a.gbase[g] = vals;
      continue;
    }
    a.gentry[g] = pos;
    a.gbase[g] = vals;
    if (pos >= ge) continue;
    const u64 rel = pos - gb;
    if (rel >= kStrEntries) {
      atomicOr(&a.status[0], 1u);
      return;
    }
    const u32 x = a.gexit[g * kStrEntries + rel];
    if (x == 0xffffffffu) {
      atomicOr(&a.status[0], 1u);
      return;
    }
    vals += a.ghops[g * kStrEntries + rel];
    pos = ge + x;
  }
  *reinterpret_cast<u64*>(a.status + 2) = vals;
}

// entry offset and first value index of every chunk, one thread per group
__global__ void __launch_bounds__(kBlock) k_str_chunk_entries(StrScanArgs a) {
  const u64 ngroups = (a.nchunks + kStrGroup - 1) / kStrGroup;
  const u64 g = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ngroups) return;
  const u64 c0 = g * kStrGroup;
  const u64 nc = a.nchunks - c0 < kStrGroup ? a.nchunks - c0 : kStrGroup;
  u64 pos = a.gentry[g], vals = a.gbase[g];
  for (u64 j = 0; j < nc; ++j) {
    const u64 c = c0 + j;
    const u64 cb = c * kStrChunk;
    a.cbase[c] = vals;
    if (pos == ~0ull || pos >= cb + kStrChunk || vals >= a.nvalues) {
      a.centry[c] = (u16) kStrNone;
      continue;
    }
    const u64 rel = pos - cb;
    a.centry[c] = (u16) rel;
    if (rel >= kStrEntries) {  // (the group table was OVER for this entry: not reached)
      atomicOr(&a.status[0], 1u);
      return;
    }
    const u32 x = a.exits[c * kStrEntries + rel];
    if (x == kStrNone) {
      atomicOr(&a.status[0], 1u);
      return;
    }
    vals += a.hops[c * kStrEntries + rel];
    pos = cb + kStrChunk + x;
  }
}

// marks the value starts of a chunk from its true entry and emits (len << 40 | pos)
__global__ void __launch_bounds__(1024) k_str_emit(StrScanArgs a) {
  constexpr u32 kLevels = 12;
  __shared__ __attribute__((aligned(16))) u8 bytes[kStrChunk + 16];
  __shared__ u16 L[kLevels][kStrChunk];
  __shared__ u32 markw[kStrChunk / 32];
  __shared__ u32 wpre[kStrChunk / 32];
  for (u64 c = blockIdx.x; c < a.nchunks; c += gridDim.x) {
    const u32 entry = a.centry[c];
    const u64 base = a.cbase[c];
    if (entry == kStrNone || base >= a.nvalues) continue;  // (uniform per workgroup)
    str_stage_chunk(a, c, bytes, 1024);
    if (threadIdx.x < kStrChunk / 32) markw[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < kStrChunk / 1024; ++k) {
      const u32 p = threadIdx.x + k * 1024;
      L[0][p] = (u16) str_next(bytes, p);
    }
    __syncthreads();
    u32 nlev = 1;
    for (u32 lv = 1; lv < kLevels; ++lv) {
      int any = 0;
#pragma unroll
      for (u32 k = 0; k < kStrChunk / 1024; ++k) {
        const u32 p = threadIdx.x + k * 1024;
        u32 j = L[lv - 1][p];
        if (j < kStrChunk) {
          j = L[lv - 1][j];
          any |= j < kStrChunk;
        }
        L[lv][p] = (u16) j;
      }
      nlev = lv + 1;
      if (!__syncthreads_or(any)) break;
    }
    if (threadIdx.x == 0) markw[entry >> 5] = 1u << (entry & 31);
    __syncthreads();
    // reachability doubling, top level first: after level k every position at a
    // multiple of 2^k values from the entry is marked (bits set meanwhile only add
    // further positions of the same path)
    for (int lv = (int) nlev - 1; lv >= 0; --lv) {
#pragma unroll
      for (u32 k = 0; k < kStrChunk / 1024; ++k) {
        const u32 p = threadIdx.x + k * 1024;
        if ((markw[p >> 5] >> (p & 31)) & 1) {
          const u32 j = L[lv][p];
          if (j < kStrChunk) atomicOr(&markw[j >> 5], 1u << (j & 31));
        }
      }
      __syncthreads();
    }
    // ranks: exclusive prefix of the word popcounts (128 words: two waves)
    if (threadIdx.x < kStrChunk / 32) {
      const u32 v = __popc(markw[threadIdx.x]);
      u32 incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const u32 t = __shfl_up(incl, d, 64);
        if ((int) (threadIdx.x & 63) >= d) incl += t;
      }
      wpre[threadIdx.x] = incl - v;
    }
    __syncthreads();
    const u32 first_wave_total = wpre[63] + __popc(markw[63]);
#pragma unroll
    for (u32 k = 0; k < kStrChunk / 1024; ++k) {
      const u32 p = threadIdx.x + k * 1024;
      const u32 w = markw[p >> 5];
      if (!((w >> (p & 31)) & 1)) continue;
      const u32 rank = wpre[p >> 5] + ((p >> 5) >= 64 ? first_wave_total : 0) +
                       __popc(w & ((1u << (p & 31)) - 1u));
      const u64 idx = base + rank;
      if (idx >= a.nvalues) continue;
      // the full header (a length is a u32: at most 5 bytes)
      u64 len = 0;
      u32 hdr = 0;
      bool bad = true;
      for (u32 i = 0; i < 5; ++i) {
        const u32 b = bytes[p + i];
        len |= (u64) (b & 0x7f) << (7 * i);
        hdr = i + 1;
        if (!(b & 0x80)) {
          bad = false;
          break;
        }
      }
      const u64 pos = c * kStrChunk + p + hdr;
      if (bad || pos + len > a.nbytes || (len >> 24) || (pos >> 40)) {
        atomicOr(&a.status[0], 2u);
        len = 0;
      }
      a.strval[idx] = (len << 40) | (pos & 0xffffffffffull);
    }
    __syncthreads();
  }
}

// fallback for streams whose values outrun the tables (strings of a kilobyte and
// more at chunk borders): one thread walks the stream and writes the chunk entries
__global__ void k_str_walk_serial(StrScanArgs a) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  u64 pos = 0, v = 0, filled = 0;
  while (v < a.nvalues && pos < a.nbytes) {
    const u64 c = pos / kStrChunk;
    for (; filled < c; ++filled) {
      a.centry[filled] = (u16) kStrNone;
      a.cbase[filled] = v;
    }
    if (c >= filled) {
      a.centry[c] = (u16) (pos - c * kStrChunk);
      a.cbase[c] = v;
      filled = c + 1;
    }
    u64 len = 0;
    u32 hdr = 0;
    bool bad = true;
    for (u32 i = 0; i < 5 && pos + i < a.nbytes; ++i) {
      const u32 b = vbyte(a.image, (const u64*) a.pages, pos + i);
      len |= (u64) (b & 0x7f) << (7 * i);
      hdr = i + 1;
      if (!(b & 0x80)) {
        bad = false;
        break;
      }
    }
    if (bad) {
      atomicOr(&a.status[0], 2u);
      break;
    }
    pos += hdr + len;
    ++v;
  }
  for (; filled < a.nchunks; ++filled) {
    a.centry[filled] = (u16) kStrNone;
    a.cbase[filled] = v;
  }
  *reinterpret_cast<u64*>(a.status + 2) = v;
}

__global__ void k_copy_strings(const u8* image, const u64* pages, const u64* strpos_list,
                               const u64* out_offsets, u64 n, u8* out) {
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    const u64 sp = strpos_list[i];
    const u64 off = sp & 0xffffffffffull;
    const u32 len = (u32) (sp >> 40);
    u8* dst = out + out_offsets[i];
    for (u32 k = 0; k < len; ++k) dst[k] = vbyte(image, pages, off + k);
  }
}

// ---- ORDER BY .. LIMIT ----------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_order_keys(OrderKeyArgs a) {
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < a.n;
       i += (u64) gridDim.x * blockDim.x) {
    const u64* rec = (const u64*) a.records + i * a.record_words;
    u64 v = rec[a.word];
    if (a.from_ident) {
      // kind 2 = NULL key: payload 0 (the comparators ignore tags)
      if (rec[0] == 2) v = 0;
    } else if (a.count_word >= 0) {
      const u64 cnt = rec[a.count_word];
      if (cnt == 0) {
        v = 0;
      } else if (a.is_mean) {
        v = evql_f64_bits(evql_as_f64(v) / (double) cnt);
      }
    }
    u64 key;
    if (a.type == 0) {
      key = v;
    } else if (a.type == 1) {
      key = v ^ 0x8000000000000000ull;
    } else {
      if ((v << 1) == 0) v = 0;  // -0.0 == +0.0 for cmp_float64
      key = (v >> 63) ? ~v : (v | 0x8000000000000000ull);
    }
    a.keys[i] = a.descending ? ~key : key;
  }
}

__global__ void __launch_bounds__(kBlock) k_radix_hist(const u64* keys, u64 n, u64 hi_mask,
                                                       u64 hi_value, u32 shift, u64* hist) {
  __shared__ u32 h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    const u64 k = keys[i];
    if ((k & hi_mask) == hi_value) atomicAdd(&h[(k >> shift) & 255], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd((unsigned long long*) &hist[threadIdx.x], (unsigned long long) h[threadIdx.x]);
}

__global__ void __launch_bounds__(kBlock) k_order_collect(const u64* keys, u64 n, u64 threshold,
                                                          u64 max_eq, u64* out_idx, u64* counters) {
  // counters: [0] lt taken, [1] eq taken, [2] output cursor
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    const u64 k = keys[i];
    bool take = k < threshold;
    if (take) {
      atomicAdd((unsigned long long*) &counters[0], 1ull);
    } else if (k == threshold && *(volatile u64*) &counters[1] < max_eq) {
      // (the plain read keeps millions of surplus ties off the atomic)
      take = atomicAdd((unsigned long long*) &counters[1], 1ull) < max_eq;
    }
    if (take) out_idx[atomicAdd((unsigned long long*) &counters[2], 1ull)] = i;
  }
}

__global__ void __launch_bounds__(kBlock) k_gather_records(const u64* records, u32 rw,
                                                           const u64* idx, u64 m, u64* out) {
  const u64 total = m * rw;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (u64) gridDim.x * blockDim.x) {
    out[i] = records[idx[i / rw] * rw + i % rw];
  }
}

// ---- LSM row filters ----------------------------------------------------------------
// The reference walks the chain sequentially with a std::set of updated ids
// (partition_cursor.cc:160-195): a row is dropped when it is skipped or when an
// EARLIER kept row with the same id was an update.  Equivalently, per id, rows are
// kept up to and including the first non-skipped update in scan order -- which two
// data-parallel passes compute with a hash table of minimum positions.
__device__ __forceinline__ bool lsm_row(const LsmArgs& a, u64 r, u64* id, bool* upd, bool* bad) {
  bool skip = false;
  if (a.has_skip) skip = rt_column_value(a.image, a.skip, r) != 0;
  if (a.arena_skip) skip = a.arena_skip[r] != 0;
  const u64 sp = a.id_pos[r];
  const u64 pos = sp & 0xFFFFFFFFFFull;
  const u32 len = (u32) (sp >> 40);
  *bad = len != 20;  // SHA1Hash(data, size) raises otherwise (util/SHA1.cc:79-85)
  id[0] = id[1] = id[2] = 0;
  for (u32 k = 0; k < 20 && k < len; ++k) {
    id[k >> 3] |= (u64) vbyte(a.image, (const u64*) a.id.pages, pos + k) << (8 * (k & 7));
  }
  // the all-ones word marks a free slot
  if (id[0] == EVQL_EMPTY) id[0] = EVQL_EMPTY - 1;
  if (id[1] == EVQL_EMPTY) id[1] = EVQL_EMPTY - 1;
  *upd = rt_column_value(a.image, a.is_update, r) != 0;
  return skip;
}

__global__ void __launch_bounds__(kBlock) k_lsm_insert(LsmArgs a) {
  const u64 mask = a.cap - 1;
  for (u64 r = (u64) blockIdx.x * blockDim.x + threadIdx.x; r < a.nrows;
       r += (u64) gridDim.x * blockDim.x) {
    u64 id[3];
    bool upd, bad;
    const bool skip = lsm_row(a, r, id, &upd, &bad);
    if (bad) {
      atomicAdd((unsigned long long*) &a.counters[1], 1ull);
      continue;
    }
    if (skip || !upd) continue;
    // (the id set is no longer empty: what the next table's "needs a filter" test reads)
    if (__hip_atomic_load(&a.counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      atomicOr((unsigned long long*) &a.counters[2], 1ull);
    }
    u64 h = evql_mix64(id[0] ^ evql_mix64(id[1] + id[2])) & mask;
    for (u64 probe = 0; probe < a.cap; ++probe, h = (h + 1) & mask) {
      u64 prev = atomicCAS((unsigned long long*) &a.tab[h], EVQL_EMPTY, id[0]);
      if (prev != EVQL_EMPTY && prev != id[0]) continue;
      prev = atomicCAS((unsigned long long*) &a.tab[a.cap + h], EVQL_EMPTY, id[1]);
      if (prev != EVQL_EMPTY && prev != id[1]) continue;
      prev = atomicCAS((unsigned long long*) &a.tab[2 * a.cap + h], EVQL_EMPTY, id[2]);
      if (prev != EVQL_EMPTY && prev != id[2]) continue;
      atomicMin((unsigned long long*) &a.tab[3 * a.cap + h], a.pos0 + r);
      break;
    }
  }
}

__global__ void __launch_bounds__(kBlock) k_lsm_filter(LsmArgs a) {
  const u64 mask = a.cap - 1;
  const u64 nwords = (a.nrows + 63) / 64;
  const u64 wave = ((u64) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const u64 nwaves = ((u64) gridDim.x * blockDim.x) >> 6;
  const u32 lane = threadIdx.x & 63;
  for (u64 w = wave; w < nwords; w += nwaves) {
    const u64 r = w * 64 + lane;
    bool keep = false;
    if (r < a.nrows) {
      u64 id[3];
      bool upd, bad;
      const bool skip = lsm_row(a, r, id, &upd, &bad);
      keep = !skip && !bad;
      if (keep) {
        u64 h = evql_mix64(id[0] ^ evql_mix64(id[1] + id[2])) & mask;
        for (u64 probe = 0; probe < a.cap; ++probe, h = (h + 1) & mask) {
          const u64 k0 = a.tab[h];
          if (k0 == EVQL_EMPTY) break;
          if (k0 == id[0] && a.tab[a.cap + h] == id[1] && a.tab[2 * a.cap + h] == id[2]) {
            keep = a.pos0 + r <= a.tab[3 * a.cap + h];
            break;
          }
        }
      }
    }
    const u64 m = __ballot(keep);
    if (lane == 0) {
      a.bits[w] = m;
      if (m) atomicAdd((unsigned long long*) &a.counters[0], (unsigned long long) __popcll(m));
    }
  }
}

// ---- nested scans over sibling repeated groups ---------------------------------------------
__global__ void __launch_bounds__(kBlock) k_record_starts(const u8* levels, const u64* tile_offsets,
                                                          u64 nslots, u64* starts, u64 max_starts) {
  const u64 tile = blockIdx.x;
  const u64 s0 = tile * kDecodeTile + (u64) threadIdx.x * 8;
  u32 cnt = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) cnt += (s0 + j < nslots && levels[s0 + j] == 0) ? 1u : 0u;
  u32 total;
  const u32 ex = block_excl_scan(cnt, &total);
  u64 k = tile_offsets[tile] + ex;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (s0 + j < nslots && levels[s0 + j] == 0) {
      if (k < max_starts) starts[k] = s0 + j;  // (zero padding behind the last record)
      ++k;
    }
  }
}

__global__ void __launch_bounds__(kBlock) k_zip_rows(const ZipArgs* ap, u64 nrec, int pass) {
  const ZipArgs& a = *ap;
  for (u64 rec = (u64) blockIdx.x * blockDim.x + threadIdx.x; rec < nrec;
       rec += (u64) gridDim.x * blockDim.x) {
    u64 cursor[kMaxZipCols], end[kMaxZipCols], cur[kMaxZipCols];
    for (u32 c = 0; c < a.ncols; ++c) {
      cursor[c] = a.starts[c] ? a.starts[c][rec] : rec;
      end[c] = a.starts[c] ? a.starts[c][rec + 1] : rec + 1;
      cur[c] = EVQL_EMPTY;
    }
    u64 row = pass ? a.rows[rec] : 0, nrows = 0;
    u32 L = 0;
    // (a record has at least one slot in every column, so the loop emits >= 1 row; the
    // bound keeps a damaged level stream from spinning)
    for (u64 guard = 0; guard < (1ull << 32); ++guard) {
      u32 next_level = 0;
      for (u32 c = 0; c < a.ncols; ++c) {
        const bool more = cursor[c] < end[c];
        const u32 nr = (more && a.levels[c] && cursor[c] != (a.starts[c] ? a.starts[c][rec] : rec))
                           ? a.levels[c][cursor[c]] : 0u;
        if (more && nr >= L) cur[c] = cursor[c]++;
        const u32 nr2 = (cursor[c] < end[c] && a.levels[c]) ? a.levels[c][cursor[c]] : 0u;
        next_level = nr2 > next_level ? nr2 : next_level;
      }
      if (pass) {
        for (u32 c = 0; c < a.ncols; ++c) a.idx[c][row] = cur[c];
      }
      ++row;
      ++nrows;
      // cur_select_level_ = cur_fetch_level_; columns at or below it are reset (:511-515)
      for (u32 c = 0; c < a.ncols; ++c) {
        if (a.rmax[c] >= next_level) cur[c] = EVQL_EMPTY;
      }
      L = next_level;
      if (L == 0) break;
    }
    if (!pass) a.rows[rec] = nrows;
  }
}

__global__ void __launch_bounds__(kBlock) k_zip_gather(const u64* vals, const u64* idx, u64 n, u64* out) {
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    const u64 k = idx[i];
    out[i] = k == EVQL_EMPTY ? 0ull : vals[k];
  }
}

// ---- result emission on the device ---------------------------------------------------
constexpr u8 kStagNull = 1;  // STAG_NULL, sql/svalue.h:52-56
__device__ __forceinline__ void emit_store9(u8* out, u64 i, u64 bits, u8 tag) {
  u8* p = out + i * 9;
#pragma unroll
  for (int b = 0; b < 8; ++b) p[b] = (u8) (bits >> (8 * b));
  p[8] = tag;
}

__global__ void __launch_bounds__(kBlock) k_extract_word(const u64* records, u64 n, u32 rw, u32 word,
                                                         u64* out) {
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    out[i] = records[i * rw + word];
  }
}

__global__ void __launch_bounds__(kBlock) k_emit_fixed(const EmitArgs* ap, u64 n) {
  const EmitArgs& a = *ap;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    const u64* rec = (const u64*) a.records + i * a.rw;
    for (u32 c = 0; c < a.ncols; ++c) {
      const EmitCol& e = a.col[c];
      if (e.elem == 0) continue;  // strings: k_emit_str_*
      u64 bits = 0;
      u8 tag = 0;
      if (e.kind == 0) {
        if (rec[0] == 2) tag = kStagNull; else bits = rec[1];
      } else if (e.kind == 1) {
        bits = rec[e.word];
        if (e.count_word >= 0) {
          const u64 cnt = rec[e.count_word];
          if (cnt == 0) {
            bits = 0;
            tag = kStagNull;
          } else if (e.is_mean) {
            const double m = evql_as_f64(bits) / (double) cnt;
            bits = evql_f64_bits(m);
          }
        }
      } else {
        const u64 raw = ((const u64*) a.first_vals)[(u64) e.src * n + i];
        tag = a.first_tags[(u64) e.src * n + i] & 1 ? kStagNull : 0;
        bits = e.to_float ? evql_f64_bits((double) raw) : raw;
      }
      if (e.elem == 2) {  // BOOL: value byte, tag byte
        e.out[i * 2] = (u8) (bits != 0);
        e.out[i * 2 + 1] = tag;
      } else {
        emit_store9(e.out, i, bits, tag);
      }
    }
  }
}

// STRING elements: u32 length, bytes, tag (a NULL string is length 0 with STAG_NULL)
__global__ void __launch_bounds__(kBlock) k_emit_str_sizes(const EmitArgs* ap, u32 c, u64 n, u64* sizes) {
  const EmitArgs& a = *ap;
  const EmitCol& e = a.col[c];
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    const bool null = a.first_tags[(u64) e.src * n + i] & 1;
    const u64 sp = ((const u64*) a.first_vals)[(u64) e.src * n + i];
    sizes[i] = 5 + (null ? 0 : (sp >> 40));
  }
}

__global__ void __launch_bounds__(kBlock) k_emit_str_bytes(const EmitArgs* ap, u32 c, u64 n) {
  const EmitArgs& a = *ap;
  const EmitCol& e = a.col[c];
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    const bool null = a.first_tags[(u64) e.src * n + i] & 1;
    const u64 sp = ((const u64*) a.first_vals)[(u64) e.src * n + i];
    const u32 len = null ? 0u : (u32) (sp >> 40);
    const u64 off = sp & kStrOffMask;
    u8* p = e.out + ((const u64*) e.offsets)[i];
    p[0] = (u8) len; p[1] = (u8) (len >> 8); p[2] = (u8) (len >> 16); p[3] = (u8) (len >> 24);
    for (u32 k = 0; k < len; ++k) p[4 + k] = vbyte_fwd(a.image, (const u64*) e.pages, off + k);
    p[4 + len] = null ? kStagNull : 0;
  }
}

// ---- string dictionaries (string_dict.cc) ---------------------------------------------
// records of the dictionary's GROUP BY over the column's 64-bit string hash:
// [kind, hash, first row, count]; record i becomes code i
__global__ void __launch_bounds__(kBlock) k_dict_insert(DictArgs a) {
  const u64 mask = a.cap - 1;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < a.ncodes;
       i += (u64) gridDim.x * blockDim.x) {
    const u64* rec = (const u64*) a.records + i * 4;
    const u64 kind = rec[0], h = rec[1];
    a.entries[i * 3] = kind == 1 ? EVQL_EMPTY : h;
    a.entries[i * 3 + 1] = rec[2];
    a.entries[i * 3 + 2] = kind == 2 ? 1ull : 0ull;
    if (kind == 1) { a.special[0] = i; continue; }  // the hash value 2^64-1
    if (kind == 2) { a.special[1] = i; continue; }  // the NULL key
    u64 s = evql_mix64(h) & mask;
    for (u64 probe = 0; probe < a.cap; ++probe, s = (s + 1) & mask) {
      // (hashes are distinct: one record per group)
      if (atomicCAS((unsigned long long*) &a.tab[s], EVQL_EMPTY, h) == EVQL_EMPTY) {
        a.tab[a.cap + s] = i;
        break;
      }
    }
  }
}

// code of every row + the proof that the dictionary is EXACT: the bytes of the row's
// string equal the bytes of its code's representative (the group's first row)
__global__ void __launch_bounds__(kBlock) k_dict_assign(DictArgs a) {
  const u64 mask = a.cap - 1;
  for (u64 r = (u64) blockIdx.x * blockDim.x + threadIdx.x; r < a.nrows;
       r += (u64) gridDim.x * blockDim.x) {
    u64 code = EVQL_EMPTY;
    const bool null = a.tags && (a.tags[r] & 1);
    const u64 h = a.hashes[r];
    if (null) {
      code = a.special[1];
    } else if (h == EVQL_EMPTY) {
      code = a.special[0];
    } else {
      u64 s = evql_mix64(h) & mask;
      for (u64 probe = 0; probe < a.cap; ++probe, s = (s + 1) & mask) {
        const u64 k = a.tab[s];
        if (k == EVQL_EMPTY) break;
        if (k == h) {
          code = a.tab[a.cap + s];
          break;
        }
      }
    }
    if (code >= a.ncodes) {
      atomicOr(&a.status[0], 1u);
      continue;
    }
    if (!null) {
      const u64 rep = a.entries[code * 3 + 1];
      if (rep != r) {
        const u64 sp = a.strpos[r], sq = a.strpos[rep];
        bool same = (sp >> 40) == (sq >> 40) && (a.entries[code * 3 + 2] & 1) == 0;
        const u32 len = (u32) (sp >> 40);
        const u64 o1 = sp & kStrOffMask, o2 = sq & kStrOffMask;
        for (u32 k = 0; same && k < len; ++k) {
          same = vbyte_fwd(a.image, (const u64*) a.pages, o1 + k) ==
                 vbyte_fwd(a.image, (const u64*) a.pages, o2 + k);
        }
        if (!same) atomicOr(&a.status[0], 2u);  // two strings, one 64-bit hash
      }
    }
    a.codes[r] = (u32) code;
  }
}

// group records keyed by dictionary code [kind, code, states...] -> the records of the
// same plan with a hashed string key [0, ident, ident 2, first row, states...]
// (evql_ident_add: the words the generated row function computes for that string)
__global__ void __launch_bounds__(kBlock) k_dict_records(const u64* in, u64 n, u32 in_words,
                                                         const u64* entries, u64* out) {
  const u32 ow = in_words + 2;
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    const u64* rec = in + i * in_words;
    u64* o = out + i * ow;
    const u64* e = entries + rec[1] * 3;
    u64 ident = EVQL_IDENT_SEED1, ident2 = EVQL_IDENT_SEED2;
    evql_ident_add(ident, ident2, e[0], (u32) e[2]);
    o[0] = 0;
    o[1] = evql_ident_word(ident);
    o[2] = evql_ident_word(ident2);
    o[3] = e[1];
    for (u32 w = 2; w < in_words; ++w) o[w + 2] = rec[w];
  }
}

// ---- synthetic table -----------------------------------------------------------------
// one thread generates 128 consecutive rows (= one bit-packed block)
__global__ void __launch_bounds__(kBlock) k_synth(const SynthArgs* ap, u64 nchunks) {
  const SynthArgs& a = *ap;
  const u64 chunk = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (chunk >= nchunks) return;
  const u64 r0 = chunk * 128;
  // x = M^(r0) seed by binary decomposition over the precomputed M^(2^j)
  u64 x = a.seed;
  for (int j = 0; j < 48; ++j) {
    if (!((r0 >> j) & 1)) continue;
    u64 y = 0;
    const u64* m = (const u64*) a.jump[j];
    for (int bit = 0; bit < 64; ++bit) {
      if ((x >> bit) & 1) y ^= m[bit];
    }
    x = y;
  }
  const u32 kb = a.k_bits;
  for (u32 i = 0; i < 128; ++i) {
    const u64 r = r0 + i;
    if (r >= a.num_rows) break;
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    if (a.columns & 1u) {
      const u64 k = x % a.k_mod;
      if (kb == 0) {
        *reinterpret_cast<u64*>(a.image + a.off_k + r * 8) = k;
      } else {
        // libsimdcomp layout; the image is zero-initialised
        u32* W = reinterpret_cast<u32*>(a.image + a.off_k + 4 + chunk * (u64) (16 * kb));
        const u32 l = i & 3u, kk = i >> 2, p = kk * kb, w = p >> 5, s = p & 31u;
        atomicOr(&W[4 * w + l], (u32) (k << s));
        if (s + kb > 32) atomicOr(&W[4 * (w + 1) + l], (u32) (k >> (32 - s)));
      }
    }
    if (a.columns & 2u) *reinterpret_cast<u64*>(a.image + a.off_a + r * 8) = (x >> 8) & 0xffffull;
    if (a.columns & 4u) *reinterpret_cast<u64*>(a.image + a.off_b + r * 8) = (x >> 24) & 0xffffull;
    if (a.columns & 8u) {
      *reinterpret_cast<double*>(a.image + a.off_v + r * 8) = (double) (x >> 40) / 1024.0;
    }
    if (a.columns & 16u) *reinterpret_cast<u64*>(a.image + a.off_u + r * 8) = x % a.u_mod;
  }
}

// ---- partitioned aggregation: per-bucket prefix over workgroups ---------------------
// counts[b][w] (tuples of workgroup w in bucket b) -> exclusive prefix inside the
// bucket, totals[b] = size of bucket b
__global__ void __launch_bounds__(kBlock) k_part_scan(u32* counts, u64 nwg, u64* totals) {
  const u64 b = blockIdx.x;
  u32 carry = 0;
  for (u64 base = 0; base < nwg; base += kBlock) {
    const u64 w = base + threadIdx.x;
    const u32 v = w < nwg ? counts[b * nwg + w] : 0;
    u32 total;
    const u32 ex = block_excl_scan(v, &total);
    if (w < nwg) counts[b * nwg + w] = carry + ex;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[b] = carry;
}

// ---- nested (Dremel) scans ---------------------------------------------------------
// level stream (bit-packed, width `bits`) -> one byte per slot, plus per-tile
// counts of slots with level <= thr[c] for up to 4 thresholds (thr = 255 => skip)
__global__ void __launch_bounds__(kBlock) k_level_decode(LevelDecodeArgs a) {
  const u64 tile = blockIdx.x;
  const u64 s0 = tile * kDecodeTile + (u64) threadIdx.x * 8;
  u32 cnt[4] = {0, 0, 0, 0};
  u64 packed = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const u64 s = s0 + j;
    u32 lv = 0xff;
    if (s < a.nslots) {
      lv = evql_bitpacked_rt(a.image, (const u64*) a.pages, a.bits, s);
#pragma unroll
      for (int c = 0; c < 4; ++c) cnt[c] += (lv <= a.thr[c]) ? 1 : 0;
    }
    packed |= (u64) (lv & 0xff) << (8 * j);
  }
  *reinterpret_cast<u64*>(a.levels + s0) = packed;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (a.counts[c] == nullptr) continue;
    u32 total;
    block_excl_scan(cnt[c], &total);
    if (threadIdx.x == 0) a.counts[c][tile] = total;
  }
}

}  // namespace

hipError_t launch_level_decode(const LevelDecodeArgs& a, hipStream_t s) {
  const u64 ntiles = (a.nslots + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_level_decode, dim3((unsigned) ntiles), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

namespace {

// first slot index at which the inclusive count of (level <= thr) exceeds
// `target` (i.e. the start of record number `target`), given the scanned tile
// offsets; writes the stream's slot count when the target is never reached
__global__ void __launch_bounds__(kBlock) k_find_nth(const u8* levels, const u64* tile_offsets,
                                                     u64 ntiles, u64 nslots, u32 thr, u64 target,
                                                     u64* out) {
  const u64 tile = blockIdx.x;
  const u64 base = tile_offsets[tile];
  const u64 next = tile + 1 < ntiles ? tile_offsets[tile + 1] : ~0ull;
  if (tile == 0 && threadIdx.x == 0 && target == 0) {
    // degenerate: zero records
  }
  if (!(base <= target && target < next)) return;
  // this tile contains the target-th qualifying slot (0-based): serial scan by
  // one thread (2048 bytes)
  if (threadIdx.x != 0) return;
  u64 seen = base;
  for (u32 i = 0; i < kDecodeTile; ++i) {
    const u64 s = tile * kDecodeTile + i;
    if (s >= nslots) break;
    if (levels[s] <= thr) {
      if (seen == target) {
        *out = s;
        return;
      }
      ++seen;
    }
  }
}

// flat[j] = vals[(inclusive count of leaf slots i <= j with level <= thr) - 1]
__global__ void __launch_bounds__(kBlock) k_flatten_parent(const u8* leaf_levels,
                                                           const u64* tile_offsets, u32 thr,
                                                           u64 nflat, const u64* vals, u64* flat) {
  const u64 tile = blockIdx.x;
  const u64 s0 = tile * kDecodeTile + (u64) threadIdx.x * 8;
  const u64 packed = *reinterpret_cast<const u64*>(leaf_levels + s0);
  u32 cnt = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    cnt += (((packed >> (8 * j)) & 0xff) <= thr && s0 + j < nflat) ? 1 : 0;
  }
  u32 total;
  u64 idx = tile_offsets[tile] + block_excl_scan(cnt, &total);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const u64 s = s0 + j;
    if (((packed >> (8 * j)) & 0xff) <= thr && s < nflat) ++idx;
    flat[s] = (s < nflat && idx > 0) ? vals[idx - 1] : 0;
  }
}

// ---- CSTableScan's reset of parent values behind a rejected row (CSTableScan.cc:501-512)
__global__ void __launch_bounds__(kBlock) k_level_tile_counts(const u8* levels, u32 thr, u64 n,
                                                              u64* tile_counts) {
  const u64 s0 = (u64) blockIdx.x * kDecodeTile + (u64) threadIdx.x * 8;
  const u64 packed = *reinterpret_cast<const u64*>(levels + s0);
  u32 cnt = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) cnt += (((packed >> (8 * j)) & 0xff) <= thr && s0 + j < n) ? 1 : 0;
  u32 total;
  block_excl_scan(cnt, &total);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kBlock) k_slot_keep(const u8* levels, const u64* tile_offsets,
                                                      u32 thr, u64 n, const u8* acc, u8* slot_keep) {
  const u64 s0 = (u64) blockIdx.x * kDecodeTile + (u64) threadIdx.x * 8;
  const u64 packed = *reinterpret_cast<const u64*>(levels + s0);
  u32 cnt = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) cnt += (((packed >> (8 * j)) & 0xff) <= thr && s0 + j < n) ? 1 : 0;
  u32 total;
  u64 idx = tile_offsets[blockIdx.x] + block_excl_scan(cnt, &total);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (((packed >> (8 * j)) & 0xff) <= thr && s0 + j < n) slot_keep[idx++] = acc[s0 + j];
  }
}

__global__ void __launch_bounds__(kBlock) k_mask_parent(const u8* levels, const u64* tile_offsets,
                                                        u32 thr, u64 n, const u8* slot_keep,
                                                        const u64* in, u64* out) {
  const u64 s0 = (u64) blockIdx.x * kDecodeTile + (u64) threadIdx.x * 8;
  const u64 packed = *reinterpret_cast<const u64*>(levels + s0);
  u32 cnt = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) cnt += (((packed >> (8 * j)) & 0xff) <= thr && s0 + j < n) ? 1 : 0;
  u32 total;
  u64 idx = tile_offsets[blockIdx.x] + block_excl_scan(cnt, &total);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const u64 s = s0 + j;
    const bool first = ((packed >> (8 * j)) & 0xff) <= thr && s < n;
    if (first) ++idx;
    // the first row of a slot reads the freshly fetched value; the rows behind it read
    // it only if that first row passed the WHERE
    u64 v = 0;
    if (s < n) v = (first || (idx > 0 && slot_keep[idx - 1])) ? in[s] : 0;
    out[s] = v;
  }
}

// AGGREGATE_WITHIN_RECORD_FLAT: every thread walks a window of 8 consecutive
// flattened rows (rows of one record are adjacent).  Records inside one window
// are summed in registers and stored plainly.  A record that spans windows is
// owned by the thread holding its level-0 slot: it adds the `head` partial sums
// (rows before a window's first level-0 slot) of the following threads from LDS
// and stores the total.  Only the records that cross a 2048-row tile boundary
// (<= 2 per tile) need more: the owner stores what its tile holds, every tile writes the
// sum of the rows in front of its first record start into tile_head[], and
// k_within_record_carry adds those heads to the record they belong to -- one thread per
// such record, plain loads and stores.  (Round 2 added them atomically into a zero-filled
// out[]: the fill was 0.8 GB per aggregate and step.)  Measured on MI355X, mixing plain
// stores and atomics at window granularity had cost 5.5 ms per 2.1e8 rows against 1.1 ms
// for stores only.  All arrays are padded by >= 8192 slots (padded_rows): windows and the
// peek behind the tile are readable.
__global__ void __launch_bounds__(kBlock) k_within_record(WithinRecordArgs a) {
  __shared__ u64 s_head[kBlock];
  __shared__ u8 s_hz[kBlock];
  const u32 tid = threadIdx.x;
  const u64 tile = blockIdx.x;
  const u64 s0 = tile * kDecodeTile + (u64) tid * 8;
  u64 packed = 0;        // level bytes of the window (no level stream: all 0)
  u64 idx0 = s0;         // records started before row s0
  if (a.leaf_levels) {
    packed = *reinterpret_cast<const u64*>(a.leaf_levels + s0);
    u32 cnt = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      cnt += (((packed >> (8 * j)) & 0xff) == 0 && s0 + j < a.nflat) ? 1 : 0;
    }
    u32 total;
    idx0 = a.rec_offsets[tile] + block_excl_scan(cnt, &total);
  }
  // has_zero: the window holds a level-0 slot; a window behind the last row ends
  // every chain with an empty head
  bool hz = s0 >= a.nflat;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    hz = hz || (s0 + j < a.nflat && ((packed >> (8 * j)) & 0xff) == 0);
  }
  s_hz[tid] = hz ? 1 : 0;
  for (u32 e = 0; e < a.n; ++e) {
    const u32 lmax = a.level[e];
    u64 v[8];
    if (!a.is_count[e] && a.src[e]) {
      typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
      const u64x2_t* p = reinterpret_cast<const u64x2_t*>(a.src[e] + s0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u64x2_t x = __builtin_nontemporal_load(p + j);
        v[2 * j] = x.x;
        v[2 * j + 1] = x.y;
      }
    } else {
      const u64 c = a.is_count[e] ? 1ull : a.lit[e];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = c;
    }
    u64* out = (u64*) a.out[e];
    u64 idx = idx0, acc = 0, head = 0;
    bool started = false;  // a level-0 slot was seen in this window
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (s0 + j < a.nflat) {
        const u32 lvl = (u32) ((packed >> (8 * j)) & 0xff);
        if (lvl == 0) {
          if (!started) {
            head = acc;
          } else if (idx - 1 < a.nrec) {
            out[idx - 1] = acc;  // record inside the window
          }
          acc = 0;
          ++idx;
          started = true;
        }
        if (lmax >= lvl) acc += v[j];
      }
    }
    if (!started) head = acc;
    s_head[tid] = head;
    __syncthreads();
    if (started) {
      // the record of the window's last level-0 slot: + the heads that follow
      u64 sum = acc;
      u32 t = tid + 1;
      while (t < kBlock && !s_hz[t]) sum += s_head[t++];
      if (t < kBlock) sum += s_head[t];
      // (a record that runs on into the next tile: k_within_record_carry adds the rest)
      if (idx - 1 < a.nrec) out[idx - 1] = sum;
    }
    if (tid == 0 && a.tile_head[e]) {
      // the rows in front of the tile's first record start belong to record idx0 - 1,
      // which began in an earlier tile
      u64 sum = 0;
      if (idx0 > 0 && s0 < a.nflat && (packed & 0xff) != 0) {
        u32 t = 0;
        while (t < kBlock && !s_hz[t]) sum += s_head[t++];
        if (t < kBlock) sum += s_head[t];
      }
      a.tile_head[e][tile] = sum;
    }
    __syncthreads();
  }
}

// One thread per tile that starts inside a record whose first row lies in the tile before
// it: the heads of this tile and of the tiles behind it that hold no record start at all
// are added to that record.  Every such record has exactly one thread.
__global__ void __launch_bounds__(kBlock) k_within_record_carry(WithinRecordArgs a, u64 ntiles) {
  const u64 t = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0 || t >= ntiles || t * kDecodeTile >= a.nflat) return;
  if (a.leaf_levels[t * kDecodeTile] == 0) return;             // starts with a record
  if (a.rec_offsets[t] == a.rec_offsets[t - 1]) return;        // the record began further back
  const u64 rec = a.rec_offsets[t] - 1;
  if (rec >= a.nrec) return;
  u64 last = t;
  while (last + 1 < ntiles && (last + 1) * kDecodeTile < a.nflat &&
         a.rec_offsets[last + 1] == a.rec_offsets[last]) {
    ++last;
  }
  for (u32 e = 0; e < a.n; ++e) {
    u64 sum = 0;
    for (u64 u = t; u <= last; ++u) sum += a.tile_head[e][u];
    if (sum) ((u64*) a.out[e])[rec] += sum;
  }
}

// tags (0 defined / 1 undefined) of a nested column from decoded level bytes
__global__ void __launch_bounds__(kBlock) k_defined_from_levels(const u8* dlevels, u32 dmax,
                                                                u64 nslots, u8* tags,
                                                                u64* tile_counts) {
  const u64 tile = blockIdx.x;
  const u64 s0 = tile * kDecodeTile + (u64) threadIdx.x * 8;
  const u64 packed = *reinterpret_cast<const u64*>(dlevels + s0);
  u32 cnt = 0;
  u64 out = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool def = (((packed >> (8 * j)) & 0xff) == dmax) && (s0 + j < nslots);
    cnt += def ? 1 : 0;
    out |= (u64) (def ? 0 : 1) << (8 * j);
  }
  *reinterpret_cast<u64*>(tags + s0) = out;
  u32 total;
  block_excl_scan(cnt, &total);
  if (threadIdx.x == 0) tile_counts[tile] = total;
}

inline int grid_for(u64 n, int block = kBlock, int cap = 8192) {
  u64 g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > (u64) cap) g = cap;
  return (int) g;
}

}  // namespace

hipError_t launch_table_init(const TableInitArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_table_init, dim3(grid_for(a.stride * a.nwords)), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_table_compact(const uint64_t* words, uint64_t gcap, uint64_t stride,
                                uint32_t nwords, uint64_t* out_records, uint64_t max_records,
                                uint64_t* counter, hipStream_t s) {
  if (max_records == 0) {
    // count only: no output positions needed, so no block scans -- with 1e7 groups
    // in 6.7e7 slots the scanning form spent 3.2 ms just to learn the group count
    hipLaunchKernelGGL(k_table_count, dim3(grid_for(gcap + 2, kBlock, 2048)), dim3(kBlock), 0, s,
                       (const u64*) words, (u64) gcap, nwords, (u64*) counter);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_table_compact, dim3(grid_for(gcap + 2)), dim3(kBlock), 0, s,
                     (const u64*) words, (u64) gcap, (u64) stride, nwords, (u64*) out_records,
                     (u64) max_records, (u64*) counter);
  return hipGetLastError();
}

hipError_t launch_table_merge(const MergeArgs& a, const uint64_t* records, uint64_t n,
                              hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_table_merge, dim3(grid_for(n)), dim3(kBlock), 0, s, a,
                     (const u64*) records, (u64) n);
  return hipGetLastError();
}

hipError_t launch_gather_rows(const uint8_t* image, const RtColumn* d_cols, uint32_t ncols,
                              const uint64_t* d_rows, uint64_t n, uint64_t* out_vals,
                              uint8_t* out_tags, hipStream_t s) {
  if (n == 0 || ncols == 0) return hipSuccess;
  hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(n)), dim3(kBlock), 0, s, image, d_cols, ncols,
                     (const u64*) d_rows, (u64) n, (u64*) out_vals, out_tags);
  return hipGetLastError();
}

hipError_t launch_dlevel_tags(const uint8_t* image, const uint64_t* dlevel_pages, uint32_t dbits,
                              uint32_t dmax, uint64_t nrows, uint8_t* tags,
                              uint64_t* tile_counts, hipStream_t s) {
  const u64 ntiles = (nrows + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_dlevel_tags, dim3((unsigned) ntiles), dim3(kBlock), 0, s, image,
                     (const u64*) dlevel_pages, dbits, dmax, (u64) nrows, tags,
                     (u64*) tile_counts);
  return hipGetLastError();
}

hipError_t launch_exclusive_scan(uint64_t* data, uint64_t n, uint64_t* total, hipStream_t s) {
  hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, s, (u64*) data, (u64) n,
                     (u64*) total);
  return hipGetLastError();
}

hipError_t launch_expand_nullable(const uint8_t* image, RtColumn src, const uint8_t* tags,
                                  const uint64_t* tile_offsets, uint64_t nrows,
                                  uint64_t* values, hipStream_t s) {
  const u64 ntiles = (nrows + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_expand_nullable, dim3((unsigned) ntiles), dim3(kBlock), 0, s, image, src,
                     tags, (const u64*) tile_offsets, (u64) nrows, (u64*) values);
  return hipGetLastError();
}

hipError_t launch_leb128_count(const uint8_t* image, const uint64_t* pages, uint64_t nbytes,
                               uint64_t* chunk_counts, hipStream_t s) {
  const u64 nchunks = (nbytes + kLebChunk - 1) / kLebChunk;
  if (nchunks == 0) return hipSuccess;
  hipLaunchKernelGGL(k_leb128_count, dim3((unsigned) nchunks), dim3(kBlock), 0, s, image,
                     (const u64*) pages, (u64) nbytes, (u64*) chunk_counts);
  return hipGetLastError();
}

hipError_t launch_leb128_decode(const uint8_t* image, const uint64_t* pages, uint64_t nbytes,
                                const uint64_t* chunk_offsets, uint64_t nvalues,
                                uint64_t* values, hipStream_t s) {
  const u64 nchunks = (nbytes + kLebChunk - 1) / kLebChunk;
  if (nchunks == 0) return hipSuccess;
  hipLaunchKernelGGL(k_leb128_decode, dim3((unsigned) nchunks), dim3(kBlock), 0, s, image,
                     (const u64*) pages, (u64) nbytes, (const u64*) chunk_offsets, (u64) nvalues,
                     (u64*) values);
  return hipGetLastError();
}

hipError_t launch_string_hash(const uint8_t* image, const uint64_t* pages,
                              const uint64_t* strpos, uint64_t n, uint64_t* out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_string_hash, dim3(grid_for(n)), dim3(kBlock), 0, s, image,
                     (const u64*) pages, (const u64*) strpos, (u64) n, (u64*) out);
  return hipGetLastError();
}

hipError_t launch_owner_hist(const uint64_t* records, uint64_t n, uint32_t rw, uint32_t nranks,
                             uint64_t* counts, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_owner_hist, dim3(grid_for(n, kBlock, 2048)), dim3(kBlock), 0, s,
                     (const u64*) records, (u64) n, rw, nranks, (u64*) counts);
  return hipGetLastError();
}

hipError_t launch_owner_scatter(const uint64_t* records, uint64_t n, uint32_t rw, uint32_t nranks,
                                const uint64_t* starts, uint64_t* cursors, uint64_t* out,
                                hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_owner_scatter, dim3(grid_for(n)), dim3(kBlock), 0, s, (const u64*) records,
                     (u64) n, rw, nranks, (const u64*) starts, (u64*) cursors, (u64*) out);
  return hipGetLastError();
}

hipError_t launch_pairset_export(const uint64_t* tab, uint64_t cap, uint64_t* out,
                                 uint64_t max_triples, uint64_t* counter, hipStream_t s) {
  if (cap == 0) return hipSuccess;
  hipLaunchKernelGGL(k_pairset_export, dim3(grid_for(cap, kBlock, 2048)), dim3(kBlock), 0, s,
                     (const u64*) tab, (u64) cap, (u64*) out, (u64) max_triples, (u64*) counter);
  return hipGetLastError();
}

hipError_t launch_triple_owner_hist(const uint64_t* triples, uint64_t n, uint32_t nranks,
                                    bool exact_key, uint64_t* counts, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_triple_owner_hist, dim3(grid_for(n, kBlock, 2048)), dim3(kBlock), 0, s,
                     (const u64*) triples, (u64) n, nranks, exact_key ? 1u : 0u, (u64*) counts);
  return hipGetLastError();
}

hipError_t launch_triple_owner_scatter(const uint64_t* triples, uint64_t n, uint32_t nranks,
                                       bool exact_key, const uint64_t* starts, uint64_t* cursors,
                                       uint64_t* out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_triple_owner_scatter, dim3(grid_for(n)), dim3(kBlock), 0, s,
                     (const u64*) triples, (u64) n, nranks, exact_key ? 1u : 0u, (const u64*) starts,
                     (u64*) cursors, (u64*) out);
  return hipGetLastError();
}

hipError_t launch_pairset_merge(const PairsetMergeArgs& a, const uint64_t* triples, uint64_t n,
                                hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_pairset_merge, dim3(grid_for(n)), dim3(kBlock), 0, s, a, (const u64*) triples,
                     (u64) n);
  return hipGetLastError();
}

hipError_t launch_resolve_records(const ResolveArgs& a, hipStream_t s) {
  if (a.n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_resolve_records, dim3(grid_for(a.n)), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_wire_str_sizes(const WireStrArgs& a, hipStream_t s) {
  if (a.n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_wire_str_sizes, dim3(grid_for(a.n)), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_wire_str_copy(const WireStrArgs& a, hipStream_t s) {
  if (a.n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_wire_str_copy, dim3(grid_for(a.n)), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_bucket_scatter(const BucketScatterArgs& a, hipStream_t s) {
  const u64 vtiles = (u64) a.in_regions * a.tiles_per_region;
  const size_t lds = (size_t) a.tile * a.rw * 8;
  hipLaunchKernelGGL(k_bucket_scatter, dim3(grid_for(vtiles * kBlock, kBlock, 16384)), dim3(kBlock), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_bucket_merge(const BucketMergeArgs& a, hipStream_t s) {
  const size_t lds = (size_t) (a.lds_slots + 2) * a.mw * 8 + (size_t) (a.lds_slots + 2) * 4;
  hipLaunchKernelGGL(k_bucket_merge, dim3(a.buckets < 65536 ? a.buckets : 65536), dim3(kBlock), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_table_merge_resolved(const MergeResolvedArgs& a, const uint64_t* records,
                                       uint64_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_table_merge_resolved, dim3(grid_for(n)), dim3(kBlock), 0, s, a,
                     (const u64*) records, (u64) n);
  return hipGetLastError();
}

hipError_t launch_column_abs_max(const uint8_t* image, const RtColumn& col, uint64_t n,
                                 uint32_t is_float, uint64_t* out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_column_abs_max, dim3(grid_for(n, kBlock, 4096)), dim3(kBlock), 0, s, image,
                     col, (u64) n, is_float, (u64*) out);
  return hipGetLastError();
}

hipError_t launch_max_u64(const uint64_t* values, uint64_t n, uint64_t* out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_max_u64, dim3(grid_for(n, kBlock, 4096)), dim3(kBlock), 0, s,
                     (const u64*) values, (u64) n, (u64*) out);
  return hipGetLastError();
}

hipError_t launch_str_chunk_tables(const StrScanArgs& a, hipStream_t s) {
  if (a.nchunks == 0) return hipSuccess;
  const unsigned grid = (unsigned) (a.nchunks < 2048 ? a.nchunks : 2048);
  hipLaunchKernelGGL(k_str_chunk_tables, dim3(grid), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_str_group_compose(const StrScanArgs& a, hipStream_t s) {
  const u64 ngroups = (a.nchunks + kStrGroup - 1) / kStrGroup;
  if (ngroups == 0) return hipSuccess;
  hipLaunchKernelGGL(k_str_group_compose, dim3((unsigned) ngroups), dim3(kStrEntries), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_str_chain(const StrScanArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_str_chain, dim3(1), dim3(64), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_str_chunk_entries(const StrScanArgs& a, hipStream_t s) {
  const u64 ngroups = (a.nchunks + kStrGroup - 1) / kStrGroup;
  if (ngroups == 0) return hipSuccess;
  hipLaunchKernelGGL(k_str_chunk_entries, dim3((unsigned) ((ngroups + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_str_emit(const StrScanArgs& a, hipStream_t s) {
  if (a.nchunks == 0) return hipSuccess;
  const unsigned grid = (unsigned) (a.nchunks < 1024 ? a.nchunks : 1024);
  hipLaunchKernelGGL(k_str_emit, dim3(grid), dim3(1024), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_str_walk_serial(const StrScanArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_str_walk_serial, dim3(1), dim3(64), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_copy_strings(const uint8_t* image, const uint64_t* pages,
                               const uint64_t* strpos_list, const uint64_t* out_offsets,
                               uint64_t n, uint8_t* out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_copy_strings, dim3(grid_for(n)), dim3(kBlock), 0, s, image,
                     (const u64*) pages, (const u64*) strpos_list, (const u64*) out_offsets,
                     (u64) n, out);
  return hipGetLastError();
}

hipError_t launch_part_scan(uint32_t* counts, uint64_t npart, uint64_t nwg, uint64_t* totals,
                            hipStream_t s) {
  hipLaunchKernelGGL(k_part_scan, dim3((unsigned) npart), dim3(kBlock), 0, s, counts, (u64) nwg,
                     (u64*) totals);
  return hipGetLastError();
}

hipError_t launch_find_nth(const uint8_t* levels, const uint64_t* tile_offsets, uint64_t nslots,
                           uint32_t thr, uint64_t target, uint64_t* out, hipStream_t s) {
  const u64 ntiles = (nslots + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_find_nth, dim3((unsigned) ntiles), dim3(kBlock), 0, s, levels,
                     (const u64*) tile_offsets, (u64) ntiles, (u64) nslots, thr, (u64) target,
                     (u64*) out);
  return hipGetLastError();
}

hipError_t launch_flatten_parent(const uint8_t* leaf_levels, const uint64_t* tile_offsets,
                                 uint32_t thr, uint64_t nflat, const uint64_t* vals,
                                 uint64_t* flat, hipStream_t s) {
  const u64 ntiles = (nflat + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_flatten_parent, dim3((unsigned) ntiles), dim3(kBlock), 0, s, leaf_levels,
                     (const u64*) tile_offsets, thr, (u64) nflat, (const u64*) vals, (u64*) flat);
  return hipGetLastError();
}

hipError_t launch_level_tile_counts(const uint8_t* levels, uint32_t thr, uint64_t n,
                                    uint64_t* tile_counts, hipStream_t s) {
  const u64 ntiles = (n + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_level_tile_counts, dim3((unsigned) ntiles), dim3(kBlock), 0, s, levels, thr,
                     (u64) n, (u64*) tile_counts);
  return hipGetLastError();
}

hipError_t launch_slot_keep(const uint8_t* levels, const uint64_t* tile_offsets, uint32_t thr,
                            uint64_t n, const uint8_t* acc, uint8_t* slot_keep, hipStream_t s) {
  const u64 ntiles = (n + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_slot_keep, dim3((unsigned) ntiles), dim3(kBlock), 0, s, levels,
                     (const u64*) tile_offsets, thr, (u64) n, acc, slot_keep);
  return hipGetLastError();
}

hipError_t launch_mask_parent(const uint8_t* levels, const uint64_t* tile_offsets, uint32_t thr,
                              uint64_t n, const uint8_t* slot_keep, const uint64_t* in,
                              uint64_t* out, hipStream_t s) {
  const u64 ntiles = (n + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_mask_parent, dim3((unsigned) ntiles), dim3(kBlock), 0, s, levels,
                     (const u64*) tile_offsets, thr, (u64) n, slot_keep, (const u64*) in, (u64*) out);
  return hipGetLastError();
}

hipError_t launch_within_record(const WithinRecordArgs& a, hipStream_t s) {
  const u64 ntiles = (a.nflat + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_within_record, dim3((unsigned) ntiles), dim3(kBlock), 0, s, a);
  if (a.leaf_levels && ntiles > 1) {
    hipLaunchKernelGGL(k_within_record_carry, dim3(grid_for(ntiles)), dim3(kBlock), 0, s, a, (u64) ntiles);
  }
  return hipGetLastError();
}

hipError_t launch_defined_from_levels(const uint8_t* dlevels, uint32_t dmax, uint64_t nslots,
                                      uint8_t* tags, uint64_t* tile_counts, hipStream_t s) {
  const u64 ntiles = (nslots + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_defined_from_levels, dim3((unsigned) ntiles), dim3(kBlock), 0, s, dlevels,
                     dmax, (u64) nslots, tags, (u64*) tile_counts);
  return hipGetLastError();
}

hipError_t launch_order_keys(const OrderKeyArgs& a, hipStream_t s) {
  if (a.n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_order_keys, dim3(grid_for(a.n)), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_radix_hist(const uint64_t* keys, uint64_t n, uint64_t hi_mask,
                             uint64_t hi_value, uint32_t shift, uint64_t* hist, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_radix_hist, dim3(grid_for(n, kBlock, 1024)), dim3(kBlock), 0, s,
                     (const u64*) keys, (u64) n, (u64) hi_mask, (u64) hi_value, shift, (u64*) hist);
  return hipGetLastError();
}

hipError_t launch_order_collect(const uint64_t* keys, uint64_t n, uint64_t threshold,
                                uint64_t max_eq, uint64_t* out_idx, uint64_t* counters,
                                hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_order_collect, dim3(grid_for(n)), dim3(kBlock), 0, s, (const u64*) keys,
                     (u64) n, (u64) threshold, (u64) max_eq, (u64*) out_idx, (u64*) counters);
  return hipGetLastError();
}

hipError_t launch_gather_records(const uint64_t* records, uint32_t record_words,
                                 const uint64_t* idx, uint64_t m, uint64_t* out, hipStream_t s) {
  if (m == 0) return hipSuccess;
  hipLaunchKernelGGL(k_gather_records, dim3(grid_for(m * record_words)), dim3(kBlock), 0, s,
                     (const u64*) records, record_words, (const u64*) idx, (u64) m, (u64*) out);
  return hipGetLastError();
}

hipError_t launch_lsm_insert(const LsmArgs& a, hipStream_t s) {
  if (a.nrows == 0) return hipSuccess;
  hipLaunchKernelGGL(k_lsm_insert, dim3(grid_for(a.nrows)), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_lsm_filter(const LsmArgs& a, hipStream_t s) {
  if (a.nrows == 0) return hipSuccess;
  hipLaunchKernelGGL(k_lsm_filter, dim3(grid_for(a.nrows)), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_dict_insert(const DictArgs& a, hipStream_t s) {
  if (a.ncodes == 0) return hipSuccess;
  hipLaunchKernelGGL(k_dict_insert, dim3(grid_for(a.ncodes)), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_dict_assign(const DictArgs& a, hipStream_t s) {
  if (a.nrows == 0) return hipSuccess;
  hipLaunchKernelGGL(k_dict_assign, dim3(grid_for(a.nrows)), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_dict_records(const uint64_t* in, uint64_t n, uint32_t in_words,
                               const uint64_t* entries, uint64_t* out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_dict_records, dim3(grid_for(n)), dim3(kBlock), 0, s, (const u64*) in, (u64) n,
                     in_words, (const u64*) entries, (u64*) out);
  return hipGetLastError();
}

hipError_t launch_extract_word(const uint64_t* records, uint64_t n, uint32_t rw, uint32_t word,
                               uint64_t* out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_extract_word, dim3(grid_for(n)), dim3(kBlock), 0, s, (const u64*) records, (u64) n,
                     rw, word, (u64*) out);
  return hipGetLastError();
}

hipError_t launch_emit_fixed(const EmitArgs* d_args, uint64_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_emit_fixed, dim3(grid_for(n)), dim3(kBlock), 0, s, d_args, (u64) n);
  return hipGetLastError();
}

hipError_t launch_emit_str_sizes(const EmitArgs* d_args, uint32_t c, uint64_t n, uint64_t* sizes,
                                 hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_emit_str_sizes, dim3(grid_for(n)), dim3(kBlock), 0, s, d_args, c, (u64) n,
                     (u64*) sizes);
  return hipGetLastError();
}

hipError_t launch_emit_str_bytes(const EmitArgs* d_args, uint32_t c, uint64_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_emit_str_bytes, dim3(grid_for(n)), dim3(kBlock), 0, s, d_args, c, (u64) n);
  return hipGetLastError();
}

hipError_t launch_record_starts(const uint8_t* levels, const uint64_t* tile_offsets, uint64_t nslots,
                                uint64_t* starts, uint64_t max_starts, hipStream_t s) {
  const u64 ntiles = (nslots + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_record_starts, dim3((unsigned) ntiles), dim3(kBlock), 0, s, levels,
                     (const u64*) tile_offsets, (u64) nslots, (u64*) starts, (u64) max_starts);
  return hipGetLastError();
}

hipError_t launch_zip_rows(const ZipArgs* d_args, uint64_t nrec, int pass, hipStream_t s) {
  if (nrec == 0) return hipSuccess;
  hipLaunchKernelGGL(k_zip_rows, dim3(grid_for(nrec)), dim3(kBlock), 0, s, d_args, (u64) nrec, pass);
  return hipGetLastError();
}

hipError_t launch_zip_gather(const uint64_t* vals, const uint64_t* idx, uint64_t n, uint64_t* out,
                             hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_zip_gather, dim3(grid_for(n)), dim3(kBlock), 0, s, (const u64*) vals,
                     (const u64*) idx, (u64) n, (u64*) out);
  return hipGetLastError();
}

// ---- device-side cstable writer (io/cstable/page_writer_*.cc) -----------------------
namespace {

__device__ __forceinline__ u8* wr_stream_byte(u8* image, const u64* pages, u64 pos) {
  return image + pages[pos >> 19] + (pos & 0x7ffffu);  // 512 KiB pages
}

// the NULL flags of rows r0 .. r0 + 7, one per byte (rows behind the table: NULL);
// one 8-byte load where the caller's array allows it
__device__ __forceinline__ u64 wr_null_bytes(const u8* nulls, u64 r0, u64 nrows) {
  if (r0 + 8 <= nrows && (reinterpret_cast<uintptr_t>(nulls) & 7) == 0) {
    return *reinterpret_cast<const u64*>(nulls + r0);
  }
  u64 p = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    p |= (u64) ((r0 + j < nrows && nulls[r0 + j] == 0) ? 0 : 1) << (8 * j);
  }
  return p;
}

// per-tile (2048 rows) counts of the defined rows (nulls[r] == 0)
__global__ void __launch_bounds__(kBlock) k_wr_count_defined(const u8* nulls, u64 nrows,
                                                             u64* tile_counts) {
  const u64 r0 = (u64) blockIdx.x * kDecodeTile + (u64) threadIdx.x * 8;
  const u64 nb = wr_null_bytes(nulls, r0, nrows);
  u32 cnt = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) cnt += ((nb >> (8 * j)) & 0xff) == 0 ? 1 : 0;
  u32 total;
  block_excl_scan(cnt, &total);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = total;
}

// dense[#defined rows before r] = values[r] for every defined row r
__global__ void __launch_bounds__(kBlock) k_wr_compact(const u64* values, const u8* nulls,
                                                       const u64* tile_offsets, u64 nrows,
                                                       u64* dense) {
  const u64 r0 = (u64) blockIdx.x * kDecodeTile + (u64) threadIdx.x * 8;
  const u64 nb = wr_null_bytes(nulls, r0, nrows);
  u32 cnt = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) cnt += ((nb >> (8 * j)) & 0xff) == 0 ? 1 : 0;
  u32 total;
  u64 idx = tile_offsets[blockIdx.x] + block_excl_scan(cnt, &total);
  u64 v[8];
  if (r0 + 8 <= nrows && (reinterpret_cast<uintptr_t>(values) & 15) == 0) {
    typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
    const u64x2_t* p = reinterpret_cast<const u64x2_t*>(values + r0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u64x2_t x = p[j];
      v[2 * j] = x.x;
      v[2 * j + 1] = x.y;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = r0 + j < nrows ? values[r0 + j] : 0;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (((nb >> (8 * j)) & 0xff) == 0) dense[idx++] = v[j];
  }
}

// BitPackedIntPageWriter (page_writer_bitpacked.cc:43-100): blocks of 128 values in
// libsimdcomp's 4-lane layout, 1024 blocks per page, a 4-byte max_value header in
// front of the first page.  One thread per 32-bit output word: word w of lane l
// holds bits [32w, 32w + 32) of the lane's stream of 32 `bits`-wide values.
// dense == NULL: the stream of definition levels, 1 - nulls[i].
__global__ void __launch_bounds__(kBlock) k_wr_bitpack(u8* image, const u64* pages,
                                                       const u64* dense, const u8* nulls, u64 n,
                                                       u32 bits) {
  const u64 wi = (u64) blockIdx.x * kBlock + threadIdx.x;
  const u64 nblocks = (n + 127) / 128;
  if (wi >= nblocks * 4 * bits) return;
  const u64 blk = wi / (4 * bits);
  const u32 rem = (u32) (wi % (4 * bits));
  const u32 w = rem >> 2, lane = rem & 3;
  const u64 mask = bits >= 32 ? 0xffffffffull : ((1ull << bits) - 1);
  u32 word = 0;
  for (u32 k = (32 * w) / bits; k < 32 && k * bits < 32 * w + 32; ++k) {
    const u64 i = blk * 128 + 4 * k + lane;
    u64 v = 0;
    if (i < n) v = dense ? dense[i] : (nulls[i] ? 0ull : 1ull);
    v &= mask;
    const int sh = (int) (k * bits) - (int) (32 * w);
    word |= sh >= 0 ? (u32) (v << sh) : (u32) (v >> (-sh));
  }
  const u64 page = blk / 1024, inblk = blk % 1024;
  u8* dst = image + pages[page] + (page == 0 ? 4 : 0) + inblk * 16 * bits + (u64) rem * 4;
  *reinterpret_cast<u32*>(dst) = word;
}

// UInt64PageWriter / UInt32PageWriter: value i at stream byte i * width
__global__ void __launch_bounds__(kBlock) k_wr_plain(u8* image, const u64* pages,
                                                     const u64* dense, u64 n, u32 width) {
  for (u64 i = (u64) blockIdx.x * kBlock + threadIdx.x; i < n; i += (u64) gridDim.x * kBlock) {
    u8* dst = wr_stream_byte(image, pages, i * width);
    if (width == 8) {
      *reinterpret_cast<u64*>(dst) = dense[i];
    } else {
      *reinterpret_cast<u32*>(dst) = (u32) dense[i];
    }
  }
}

__device__ __forceinline__ u32 wr_leb_len(u64 v) {
  const u32 nbits = v ? 64 - (u32) __clzll((long long) v) : 1;
  return (nbits + 6) / 7;
}

// LEB128PageWriter (page_writer_leb128.cc): bytes per 2048-value chunk ...
__global__ void __launch_bounds__(kBlock) k_wr_leb_count(const u64* dense, u64 n,
                                                         u64* chunk_bytes) {
  const u64 i0 = (u64) blockIdx.x * kDecodeTile + (u64) threadIdx.x * 8;
  u32 len = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) len += i0 + j < n ? wr_leb_len(dense[i0 + j]) : 0;
  u32 total;
  block_excl_scan(len, &total);
  if (threadIdx.x == 0) chunk_bytes[blockIdx.x] = total;
}

// ... and the bytes themselves at their stream positions (values may straddle pages)
__global__ void __launch_bounds__(kBlock) k_wr_leb_emit(u8* image, const u64* pages,
                                                        const u64* dense, u64 n,
                                                        const u64* chunk_offsets) {
  const u64 i0 = (u64) blockIdx.x * kDecodeTile + (u64) threadIdx.x * 8;
  u32 len = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) len += i0 + j < n ? wr_leb_len(dense[i0 + j]) : 0;
  // the chunk's bytes are contiguous in the stream: staged in LDS (shifted by the
  // chunk's misalignment so that LDS words and stream words coincide), then
  // stored as whole 32-bit words -- 4-aligned stream positions never straddle a
  // page (page size and page offsets are multiples of 4); only the partial words
  // at the chunk's ends go out bytewise (the neighbour chunk owns their other bytes)
  __shared__ u32 s_bytes[(kDecodeTile * 10 + 8) / 4];
  u8* lds = reinterpret_cast<u8*>(s_bytes);
  u32 total;
  const u64 chunk_off = chunk_offsets[blockIdx.x];
  const u32 lead = (u32) (chunk_off & 3);
  u32 lp = lead + block_excl_scan(len, &total);
  for (int j = 0; j < 8; ++j) {
    if (i0 + j >= n) break;
    u64 v = dense[i0 + j];
    do {
      u8 b = v & 0x7f;
      v >>= 7;
      if (v) b |= 0x80;
      lds[lp++] = b;
    } while (v);
  }
  __syncthreads();
  const u64 a0 = chunk_off - lead;             // aligned stream position of LDS byte 0
  const u32 end = lead + total;                // LDS bytes [lead, end) are valid
  for (u32 w = threadIdx.x; w * 4 < end; w += kBlock) {
    const u32 b0 = w * 4;
    if (b0 >= lead && b0 + 4 <= end) {
      *reinterpret_cast<u32*>(wr_stream_byte(image, pages, a0 + b0)) = s_bytes[w];
    } else {
      for (u32 b = b0 < lead ? lead : b0; b < b0 + 4 && b < end; ++b) {
        *wr_stream_byte(image, pages, a0 + b) = lds[b];
      }
    }
  }
}

// repeated / nested columns arrive as one (r, d, value) triple per slot: the NULL flags
// the compaction reads (d != dlevel_max) and the level streams as words for the
// bit-packer
__global__ void __launch_bounds__(kBlock) k_wr_levels(const u8* levels, u64 n, u32 dmax,
                                                      u8* nulls, u64* words) {
  for (u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (u64) gridDim.x * blockDim.x) {
    const u8 l = levels[i];
    if (nulls) nulls[i] = l != dmax ? 1 : 0;
    if (words) words[i] = l;
  }
}

// STRING_PLAIN streams (LenencStringPageWriter, page_writer_lenencstring.cc:37-69):
// `varuint length, bytes` per value.  A value is (len << 40 | offset into `heap`).
__device__ __forceinline__ u32 wr_str_size(u64 sp) {
  const u32 len = (u32) (sp >> 40);
  return len + (len < 0x80u ? 1u : (len < 0x4000u ? 2u : (len < 0x200000u ? 3u : 4u)));
}

__global__ void __launch_bounds__(kBlock) k_wr_str_count(const u64* dense, u64 n,
                                                         u64* chunk_bytes) {
  const u64 i0 = (u64) blockIdx.x * kDecodeTile + (u64) threadIdx.x * 8;
  u32 len = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) len += i0 + j < n ? wr_str_size(dense[i0 + j]) : 0;
  u32 total;
  block_excl_scan(len, &total);
  if (threadIdx.x == 0) chunk_bytes[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kBlock) k_wr_str_emit(u8* image, const u64* pages,
                                                        const u64* dense, u64 n,
                                                        const u64* chunk_offsets, const u8* heap) {
  const u64 i0 = (u64) blockIdx.x * kDecodeTile + (u64) threadIdx.x * 8;
  u32 len = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) len += i0 + j < n ? wr_str_size(dense[i0 + j]) : 0;
  u32 total;
  u64 pos = chunk_offsets[blockIdx.x] + block_excl_scan(len, &total);
  for (int j = 0; j < 8; ++j) {
    if (i0 + j >= n) break;
    const u64 sp = dense[i0 + j];
    u32 l = (u32) (sp >> 40);
    const u8* src = heap + (sp & 0xffffffffffull);
    u32 v = l;
    do {
      u8 b = v & 0x7f;
      v >>= 7;
      if (v) b |= 0x80;
      *wr_stream_byte(image, pages, pos++) = b;
    } while (v);
    for (u32 k = 0; k < l; ++k) *wr_stream_byte(image, pages, pos++) = src[k];
  }
}

}  // namespace

// ---- page placement queries (device_writer.cc: in which order the reference's page
// writers would have allocated their pages).  Few queries (one per page), one WAVE each:
// a binary search over the scanned per-tile counts, then the 2048 entries of one tile
// split over the 64 lanes (32 each: local count, wave prefix sum, the owning lane walks
// its 32 entries).
__device__ __forceinline__ u32 wave_incl_scan(u32 v, u32 lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 t = __shfl_up(v, d, 64);
    if ((int) lane >= d) v += t;
  }
  return v;
}

// select: position of the q-th (0-based) zero byte of flags[0..n)
__global__ void __launch_bounds__(kBlock) k_wr_select_zero(const u8* flags, u64 n,
                                                           const u64* tile_offsets, u64 ntiles,
                                                           const u64* queries, u64 nq, u64* out) {
  const u64 qi = ((u64) blockIdx.x * kBlock + threadIdx.x) >> 6;
  const u32 lane = threadIdx.x & 63u;
  if (qi >= nq) return;
  const u64 q = queries[qi];
  u64 lo = 0, hi = ntiles;  // last tile whose offset is <= q
  while (hi - lo > 1) {
    const u64 mid = (lo + hi) / 2;
    if (tile_offsets[mid] <= q) lo = mid; else hi = mid;
  }
  const u64 i0 = lo * kDecodeTile + (u64) lane * 32;
  u32 cnt = 0;
  for (u32 j = 0; j < 32; ++j) cnt += (i0 + j < n && flags[i0 + j] == 0) ? 1u : 0u;
  const u32 incl = wave_incl_scan(cnt, lane);
  const u64 before = tile_offsets[lo] + incl - cnt;  // zeros before this lane's entries
  u64 pos = ~0ull;
  if (q >= before && q < before + cnt) {
    u64 seen = before;
    for (u32 j = 0; j < 32; ++j) {
      if (i0 + j < n && flags[i0 + j] == 0) {
        if (seen == q) { pos = i0 + j; break; }
        ++seen;
      }
    }
  }
  // exactly one lane (or none: q beyond the last zero) holds the answer
  const u64 ball = __ballot(pos != ~0ull);
  if (ball == 0) {
    if (lane == 0) out[qi] = n;
  } else if (pos != ~0ull) {
    out[qi] = pos;
  }
}

// rank: number of zero bytes in flags[0..q] (inclusive)
__global__ void __launch_bounds__(kBlock) k_wr_rank_zero(const u8* flags, u64 n,
                                                         const u64* tile_offsets,
                                                         const u64* queries, u64 nq, u64* out) {
  const u64 qi = ((u64) blockIdx.x * kBlock + threadIdx.x) >> 6;
  const u32 lane = threadIdx.x & 63u;
  if (qi >= nq) return;
  u64 q = queries[qi];
  if (q >= n) q = n - 1;
  const u64 t = q / kDecodeTile;
  const u64 i0 = t * kDecodeTile + (u64) lane * 32;
  u32 cnt = 0;
  for (u32 j = 0; j < 32; ++j) cnt += (i0 + j <= q && flags[i0 + j] == 0) ? 1u : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
  if (lane == 0) out[qi] = tile_offsets[t] + cnt;
}

// the value whose encoding holds byte q of a LEB128 / STRING_PLAIN stream
__global__ void __launch_bounds__(kBlock) k_wr_select_byte(const u64* dense, u64 n,
                                                           const u64* chunk_offsets, u64 nchunks,
                                                           u32 is_string, const u64* queries,
                                                           u64 nq, u64* out) {
  const u64 qi = ((u64) blockIdx.x * kBlock + threadIdx.x) >> 6;
  const u32 lane = threadIdx.x & 63u;
  if (qi >= nq) return;
  const u64 q = queries[qi];
  u64 lo = 0, hi = nchunks;
  while (hi - lo > 1) {
    const u64 mid = (lo + hi) / 2;
    if (chunk_offsets[mid] <= q) lo = mid; else hi = mid;
  }
  const u64 i0 = lo * kDecodeTile + (u64) lane * 32;
  u32 bytes = 0;
  for (u32 j = 0; j < 32; ++j) {
    if (i0 + j < n) bytes += is_string ? wr_str_size(dense[i0 + j]) : wr_leb_len(dense[i0 + j]);
  }
  // (a chunk's bytes may pass 2^32 for long strings: 64-bit prefix)
  u64 incl = bytes;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 tl = __shfl_up((u32) incl, d, 64), th = __shfl_up((u32) (incl >> 32), d, 64);
    if ((int) lane >= d) incl += ((u64) th << 32) | tl;
  }
  const u64 before = chunk_offsets[lo] + incl - bytes;
  u64 idx = ~0ull;
  if (q >= before && q < before + bytes) {
    u64 pos = before;
    for (u32 j = 0; j < 32; ++j) {
      if (i0 + j >= n) break;
      const u64 len = is_string ? wr_str_size(dense[i0 + j]) : wr_leb_len(dense[i0 + j]);
      if (q < pos + len) { idx = i0 + j; break; }
      pos += len;
    }
  }
  const u64 ball = __ballot(idx != ~0ull);
  if (ball == 0) {
    if (lane == 0) out[qi] = n;
  } else if (idx != ~0ull) {
    out[qi] = idx;
  }
}

hipError_t launch_wr_select_zero(const uint8_t* flags, uint64_t n, const uint64_t* tile_offsets,
                                 const uint64_t* queries, uint64_t nq, uint64_t* out, hipStream_t s) {
  const u64 ntiles = (n + kDecodeTile - 1) / kDecodeTile;
  if (nq == 0 || ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_wr_select_zero, dim3((unsigned) ((nq * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     s, flags, (u64) n, (const u64*) tile_offsets, ntiles, (const u64*) queries,
                     (u64) nq, (u64*) out);
  return hipGetLastError();
}

hipError_t launch_wr_rank_zero(const uint8_t* flags, uint64_t n, const uint64_t* tile_offsets,
                               const uint64_t* queries, uint64_t nq, uint64_t* out, hipStream_t s) {
  if (nq == 0 || n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_wr_rank_zero, dim3((unsigned) ((nq * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     s, flags, (u64) n, (const u64*) tile_offsets, (const u64*) queries, (u64) nq,
                     (u64*) out);
  return hipGetLastError();
}

hipError_t launch_wr_select_byte(const uint64_t* dense, uint64_t n, const uint64_t* chunk_offsets,
                                 bool is_string, const uint64_t* queries, uint64_t nq,
                                 uint64_t* out, hipStream_t s) {
  const u64 nchunks = (n + kDecodeTile - 1) / kDecodeTile;
  if (nq == 0 || nchunks == 0) return hipSuccess;
  hipLaunchKernelGGL(k_wr_select_byte, dim3((unsigned) ((nq * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     s, (const u64*) dense, (u64) n, (const u64*) chunk_offsets, nchunks,
                     is_string ? 1u : 0u, (const u64*) queries, (u64) nq, (u64*) out);
  return hipGetLastError();
}

hipError_t launch_wr_count_defined(const uint8_t* nulls, uint64_t nrows, uint64_t* tile_counts,
                                   hipStream_t s) {
  const u64 ntiles = (nrows + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_wr_count_defined, dim3((unsigned) ntiles), dim3(kBlock), 0, s, nulls,
                     (u64) nrows, (u64*) tile_counts);
  return hipGetLastError();
}

hipError_t launch_wr_compact(const uint64_t* values, const uint8_t* nulls,
                             const uint64_t* tile_offsets, uint64_t nrows, uint64_t* dense,
                             hipStream_t s) {
  const u64 ntiles = (nrows + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_wr_compact, dim3((unsigned) ntiles), dim3(kBlock), 0, s,
                     (const u64*) values, nulls, (const u64*) tile_offsets, (u64) nrows,
                     (u64*) dense);
  return hipGetLastError();
}

hipError_t launch_wr_bitpack(uint8_t* image, const uint64_t* pages, const uint64_t* dense,
                             const uint8_t* nulls, uint64_t n, uint32_t bits, hipStream_t s) {
  const u64 words = (n + 127) / 128 * 4 * bits;
  if (words == 0) return hipSuccess;
  hipLaunchKernelGGL(k_wr_bitpack, dim3((unsigned) ((words + kBlock - 1) / kBlock)), dim3(kBlock),
                     0, s, image, (const u64*) pages, (const u64*) dense, nulls, (u64) n, bits);
  return hipGetLastError();
}

hipError_t launch_wr_plain(uint8_t* image, const uint64_t* pages, const uint64_t* dense,
                           uint64_t n, uint32_t width, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_wr_plain, dim3(grid_for(n, kBlock, 65536)), dim3(kBlock), 0, s, image,
                     (const u64*) pages, (const u64*) dense, (u64) n, width);
  return hipGetLastError();
}

hipError_t launch_wr_leb_count(const uint64_t* dense, uint64_t n, uint64_t* chunk_bytes,
                               hipStream_t s) {
  const u64 ntiles = (n + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_wr_leb_count, dim3((unsigned) ntiles), dim3(kBlock), 0, s,
                     (const u64*) dense, (u64) n, (u64*) chunk_bytes);
  return hipGetLastError();
}

hipError_t launch_wr_leb_emit(uint8_t* image, const uint64_t* pages, const uint64_t* dense,
                              uint64_t n, const uint64_t* chunk_offsets, hipStream_t s) {
  const u64 ntiles = (n + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_wr_leb_emit, dim3((unsigned) ntiles), dim3(kBlock), 0, s, image,
                     (const u64*) pages, (const u64*) dense, (u64) n, (const u64*) chunk_offsets);
  return hipGetLastError();
}

hipError_t launch_synth(const SynthArgs* d_args, uint64_t num_rows, hipStream_t s) {
  const u64 nchunks = (num_rows + 127) / 128;
  if (nchunks == 0) return hipSuccess;
  hipLaunchKernelGGL(k_synth, dim3((unsigned) ((nchunks + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     s, d_args, (u64) nchunks);
  return hipGetLastError();
}

hipError_t launch_wr_levels(const uint8_t* levels, uint64_t n, uint32_t dmax, uint8_t* nulls,
                            uint64_t* words, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_wr_levels, dim3(grid_for(n)), dim3(kBlock), 0, s, levels, (u64) n, dmax,
                     nulls, (u64*) words);
  return hipGetLastError();
}

hipError_t launch_wr_str_count(const uint64_t* dense, uint64_t n, uint64_t* chunk_bytes,
                               hipStream_t s) {
  const u64 ntiles = (n + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_wr_str_count, dim3((unsigned) ntiles), dim3(kBlock), 0, s,
                     (const u64*) dense, (u64) n, (u64*) chunk_bytes);
  return hipGetLastError();
}

hipError_t launch_wr_str_emit(uint8_t* image, const uint64_t* pages, const uint64_t* dense,
                              uint64_t n, const uint64_t* chunk_offsets, const uint8_t* heap,
                              hipStream_t s) {
  const u64 ntiles = (n + kDecodeTile - 1) / kDecodeTile;
  if (ntiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_wr_str_emit, dim3((unsigned) ntiles), dim3(kBlock), 0, s, image,
                     (const u64*) pages, (const u64*) dense, (u64) n, (const u64*) chunk_offsets,
                     heap);
  return hipGetLastError();
}

}  // namespace evql
