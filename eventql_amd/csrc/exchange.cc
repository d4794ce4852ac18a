// exchange.cc -- the exchange step of a GROUP BY that ran on several GPUs.
//
// The reference fans partial aggregates in over TCP: every partition's
// PartialGroupByExpression rows travel as EVQL_OP_QUERY_PARTIALAGGR frames to the
// coordinator's GroupByMergeExpression (sql/statements/select/groupby.cc:438-472,
// 528-637; server/sql/scheduler.cc:117-162).  Here a partition's groups sit in the HBM
// of the GPU that scanned it, as dense records [kind, identity, (identity 2), (first
// row), states...]; they are bucketed by owner on the device, moved GPU to GPU (RCCL
// send/recv over xGMI, or peer copies inside one process) and merged with one kernel
// launch per source rank, in rank order, into a fresh table.
#include <rccl/rccl.h>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include "runtime.h"

using namespace evql;

namespace evql {
Status query_dense_into_table(evql_query* q);
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      return Status::error(EVQL_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    }                                                                                       \
  } while (0)

// ---------------------------------------------------------------------------------------
// in-process hub: ranks are threads of one process
// ---------------------------------------------------------------------------------------
struct evql_hub {
  int nranks = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  std::vector<const uint64_t*> host_send;              // all_gather
  std::vector<const uint64_t*> dev_send;               // all_to_all: packed send buffers
  std::vector<std::vector<uint64_t>> send_counts;      // [src][dst]
  int error = EVQL_OK;                                 // first failure of any rank (sticky)
  std::string error_msg;
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t gen = generation;
    if (++arrived == nranks) {
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
  }
};

struct evql_exchange {
  evql_ctx* ctx = nullptr;
  int nranks = 1, rank = 0;
  evql_transport_t tr{};
  std::string name;
  // built-in transports
  evql_hub* hub = nullptr;
  ncclComm_t comm = nullptr;
  uint64_t* d_scratch = nullptr;  // rccl all_gather of counts
  evql_exchange_stats_t stats{};
  // device buffers of the exchange step, kept (and only ever grown) between calls:
  // a step of a running query allocates nothing
  struct Slot {
    void* p = nullptr;
    size_t cap = 0;
  };
  Slot ws[20];
  hipError_t get(int slot, size_t bytes, void** out) {
    Slot& sl = ws[slot];
    if (bytes > sl.cap) {
      if (sl.p) hipFree(sl.p);
      sl.p = nullptr;
      sl.cap = 0;
      const size_t want = bytes + bytes / 4 + 4096;
      hipError_t e = hipMalloc(&sl.p, want);
      if (e != hipSuccess) return e;
      sl.cap = want;
    }
    *out = sl.p;
    return hipSuccess;
  }
  ~evql_exchange() {
    for (auto& sl : ws) {
      if (sl.p) hipFree(sl.p);
    }
  }
};

// a typed view of a workspace slot with DevBuf's interface
template <typename T>
struct WsBuf {
  evql_exchange* x;
  int slot;
  T* p = nullptr;
  WsBuf(evql_exchange* xx, int s) : x(xx), slot(s) {}
  hipError_t alloc(size_t bytes) { return x->get(slot, bytes ? bytes : 8, reinterpret_cast<void**>(&p)); }
  operator T*() const { return p; }
};

namespace {

// Every rank reaches every barrier of a collective even after a local failure (a rank
// that returned early would leave its peers waiting on the condition variable for
// ever); the first error is kept in the hub and every rank reports it.  An exchange
// error is fatal to the hub / communicator: the caller tears it down.
int hub_fail(evql_hub* h, int code, const char* msg) {
  std::unique_lock<std::mutex> lk(h->mu);
  if (h->error == EVQL_OK) {
    h->error = code;
    h->error_msg = msg;
  }
  return code;
}

int hub_result(evql_hub* h) {
  std::unique_lock<std::mutex> lk(h->mu);
  if (h->error != EVQL_OK) return fail(h->error, h->error_msg);
  return EVQL_OK;
}

int hub_all_gather(void* user, const uint64_t* send, uint64_t n, uint64_t* recv) {
  evql_exchange* x = static_cast<evql_exchange*>(user);
  evql_hub* h = x->hub;
  h->host_send[x->rank] = send;
  h->barrier();
  for (int r = 0; r < x->nranks; ++r) memcpy(recv + uint64_t(r) * n, h->host_send[r], n * 8);
  h->barrier();
  return hub_result(h);
}

int hub_all_to_all(void* user, const uint64_t* d_send, const uint64_t* send_counts, uint64_t* d_recv,
                   const uint64_t* recv_counts, void* stream) {
  evql_exchange* x = static_cast<evql_exchange*>(user);
  evql_hub* h = x->hub;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the send buffer has to be complete before another rank's thread reads it
  if (hipStreamSynchronize(s) != hipSuccess) hub_fail(h, EVQL_EDEVICE, "hub: stream sync failed");
  h->dev_send[x->rank] = d_send;
  h->send_counts[x->rank].assign(send_counts, send_counts + x->nranks);
  h->barrier();
  uint64_t roff = 0;
  bool ok = hub_result(h) == EVQL_OK;
  for (int r = 0; ok && r < x->nranks; ++r) {
    uint64_t soff = 0;
    for (int d = 0; d < x->rank; ++d) soff += h->send_counts[r][d];
    const uint64_t cnt = h->send_counts[r][x->rank];
    if (cnt != recv_counts[r]) {
      hub_fail(h, EVQL_ERUNTIME, "hub: counts disagree");
      ok = false;
      break;
    }
    if (cnt && hipMemcpyAsync(d_recv + roff, h->dev_send[r] + soff, cnt * 8, hipMemcpyDefault, s) !=
                   hipSuccess) {
      hub_fail(h, EVQL_EDEVICE, "hub: device copy failed");
      ok = false;
      break;
    }
    roff += cnt;
  }
  if (hipStreamSynchronize(s) != hipSuccess) hub_fail(h, EVQL_EDEVICE, "hub: stream sync failed");
  h->barrier();  // nobody reuses a send buffer before everyone has copied from it
  return hub_result(h);
}

int rccl_all_gather(void* user, const uint64_t* send, uint64_t n, uint64_t* recv) {
  evql_exchange* x = static_cast<evql_exchange*>(user);
  hipStream_t s = x->ctx->stream;
  if (n > 512) return fail(EVQL_EARG, "rccl all_gather: too many words");
  uint64_t* d_send = x->d_scratch;
  uint64_t* d_recv = x->d_scratch + 512;
  if (hipMemcpyAsync(d_send, send, n * 8, hipMemcpyHostToDevice, s) != hipSuccess) {
    return fail(EVQL_EDEVICE, "rccl all_gather: copy failed");
  }
  if (ncclAllGather(d_send, d_recv, n, ncclUint64, x->comm, s) != ncclSuccess) {
    return fail(EVQL_EDEVICE, "ncclAllGather failed");
  }
  if (hipMemcpyAsync(recv, d_recv, uint64_t(x->nranks) * n * 8, hipMemcpyDeviceToHost, s) !=
          hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) {
    return fail(EVQL_EDEVICE, "rccl all_gather: copy back failed");
  }
  return EVQL_OK;
}

int rccl_all_to_all(void* user, const uint64_t* d_send, const uint64_t* send_counts, uint64_t* d_recv,
                    const uint64_t* recv_counts, void* stream) {
  evql_exchange* x = static_cast<evql_exchange*>(user);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // xGMI is point to point: one send and one receive per peer, all in flight at once
  if (ncclGroupStart() != ncclSuccess) return fail(EVQL_EDEVICE, "ncclGroupStart failed");
  ncclResult_t rc = ncclSuccess;
  bool self_ok = true;
  uint64_t soff = 0, roff = 0;
  for (int r = 0; r < x->nranks; ++r) {
    // (a failed call inside the group: the group is still closed, so that the calls
    // already queued are matched on the peers, and the error is reported afterwards)
    if (r == x->rank) {
      // this rank's own share never leaves the device: one copy on the stream instead of
      // a send / receive pair through the communicator's staging kernels
      if (send_counts[r] != recv_counts[r]) {
        self_ok = false;
      } else if (send_counts[r] &&
                 hipMemcpyAsync(d_recv + roff, d_send + soff, send_counts[r] * 8,
                                hipMemcpyDeviceToDevice, s) != hipSuccess) {
        self_ok = false;
      }
    } else {
      if (send_counts[r] && rc == ncclSuccess) {
        rc = ncclSend(d_send + soff, send_counts[r], ncclUint64, r, x->comm, s);
      }
      if (recv_counts[r] && rc == ncclSuccess) {
        rc = ncclRecv(d_recv + roff, recv_counts[r], ncclUint64, r, x->comm, s);
      }
    }
    soff += send_counts[r];
    roff += recv_counts[r];
  }
  const ncclResult_t rce = ncclGroupEnd();
  if (rc != ncclSuccess) {
    return fail(EVQL_EDEVICE, std::string("ncclSend / ncclRecv failed: ") + ncclGetErrorString(rc));
  }
  if (rce != ncclSuccess) {
    return fail(EVQL_EDEVICE, std::string("ncclGroupEnd failed: ") + ncclGetErrorString(rce));
  }
  if (!self_ok) return fail(EVQL_EDEVICE, "exchange: copy of this rank's own share failed");
  return EVQL_OK;
}

double ms_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

static const uint32_t kMergeSkip = 254;  // (rt_atomic knows no such op: the word is left alone)

// count_distinct (aggregate.cc:77-137: a std::set per group, merged by insertion): the
// stored triples of pair set `which` go to the ranks that own their groups (all ranks
// for GATHER_ALL) and are inserted into a fresh set there; every triple that is new to
// it adds 1 to state word `word` of its group in the merged table.
static Status exchange_pairset(evql_query* q, evql_exchange* x, bool by_owner, int which,
                               uint32_t word, uint64_t mcap, uint32_t mw) {
  hipStream_t s = q->ctx->stream;
  const int N = x->nranks;
  const bool exact = q->kp.key_mode == KEY_EXACT;
  uint64_t np = 0;
  uint64_t* d_cnt = q->d_counters + 6;
  HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, s));
  if (q->d_pairset[which]) {
    HIP_TRY(launch_pairset_export(q->d_pairset[which], q->pairset_cap, nullptr, 0, d_cnt, s));
  }
  HIP_TRY(hipMemcpyAsync(&np, d_cnt, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  WsBuf<uint64_t> d_tr(x, 11), d_send(x, 12), d_aux(x, 4), d_recv(x, 13);
  HIP_TRY(d_tr.alloc(std::max<uint64_t>(np, 1) * 24));
  if (np) {
    HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, s));
    HIP_TRY(launch_pairset_export(q->d_pairset[which], q->pairset_cap, d_tr, np, d_cnt, s));
  }
  std::vector<uint64_t> send_counts(N, np), starts(N + 1, 0);
  const uint64_t* d_out = d_tr;
  if (by_owner) {
    HIP_TRY(d_send.alloc(std::max<uint64_t>(np, 1) * 24));
    HIP_TRY(d_aux.alloc((3 * kMaxExchangeRanks + 4) * 8));
    uint64_t* d_counts = d_aux.p;
    uint64_t* d_starts = d_aux.p + kMaxExchangeRanks;
    uint64_t* d_cursors = d_aux.p + 2 * kMaxExchangeRanks + 2;
    HIP_TRY(hipMemsetAsync(d_aux, 0, (3 * kMaxExchangeRanks + 4) * 8, s));
    HIP_TRY(launch_triple_owner_hist(d_tr, np, uint32_t(N), exact, d_counts, s));
    HIP_TRY(hipMemcpyAsync(send_counts.data(), d_counts, N * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int r = 0; r < N; ++r) starts[r + 1] = starts[r] + send_counts[r];
    HIP_TRY(hipMemcpyAsync(d_starts, starts.data(), (N + 1) * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(launch_triple_owner_scatter(d_tr, np, uint32_t(N), exact, d_starts, d_cursors, d_send, s));
    d_out = d_send;
  } else {
    // the same triples to everybody (replicated per destination like the records)
    HIP_TRY(d_send.alloc(std::max<uint64_t>(np * N, 1) * 24));
    for (int r = 0; r < N; ++r) {
      HIP_TRY(hipMemcpyAsync(d_send.p + uint64_t(r) * np * 3, d_tr, np * 24, hipMemcpyDeviceToDevice, s));
    }
    d_out = d_send;
  }
  std::vector<uint64_t> all(uint64_t(N) * N);
  int rc = x->tr.all_gather_u64(x->tr.user, send_counts.data(), N, all.data());
  if (rc != EVQL_OK) return Status::error(rc, "exchange: all_gather of the pair counts failed");
  std::vector<uint64_t> send_words(N), recv_words(N);
  uint64_t total = 0;
  for (int r = 0; r < N; ++r) {
    const uint64_t from_r = all[uint64_t(r) * N + x->rank];
    send_words[r] = send_counts[r] * 3;
    recv_words[r] = from_r * 3;
    total += from_r;
  }
  HIP_TRY(d_recv.alloc(std::max<uint64_t>(total, 1) * 24));
  rc = x->tr.all_to_all_words(x->tr.user, d_out, send_words.data(), d_recv, recv_words.data(), s);
  if (rc != EVQL_OK) return Status::error(rc, "exchange: transfer of the count_distinct pairs failed");
  HIP_TRY(hipStreamSynchronize(s));
  x->stats.bytes_sent += (by_owner ? np : np * uint64_t(N - 1)) * 24;
  uint64_t set_cap = 1024;
  while (set_cap < total * 2) set_cap <<= 1;
  // the merged set belongs to the query: EVQL_MODE_PARTIAL rows carry the values of every
  // group's set (aggregate.cc:111-117), read back from here at emission
  if (q->d_mset[which] && q->mset_cap[which] != set_cap) {
    hipFree(q->d_mset[which]);
    q->d_mset[which] = nullptr;
  }
  if (!q->d_mset[which]) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_mset[which]), set_cap * 24));
  }
  q->mset_cap[which] = set_cap;
  uint64_t* d_set = q->d_mset[which];
  HIP_TRY(hipMemsetAsync(d_set, 0xff, set_cap * 24, s));
  PairsetMergeArgs pa{};
  pa.set = d_set;
  pa.set_cap = set_cap;
  pa.words = q->d_mtab;
  pa.gcap = mcap;
  pa.nwords = mw;
  pa.word = word;
  pa.key_mode = uint32_t(q->kp.key_mode);
  pa.status = q->d_status;
  HIP_TRY(launch_pairset_merge(pa, d_recv, total, s));
  return Status();
}

// -----------------------------------------------------------------------------------------
// large merges: split by identity hash into LDS-sized buckets, merge every bucket in the LDS
// -----------------------------------------------------------------------------------------
// Merging 1e7 received records into an HBM hash table costs 2.1 ms however few atomics a
// record needs (scattered line accesses over a > 1 GB table).  From kBucketedMergeMin
// records on, the records take two LDS-staged scatter passes (<= 256 bins each, regions
// with slack instead of a histogram pass) and one LDS merge per bucket, and the merged
// groups stay with the query as dense records (q->d_mdense).  *done = false: the shape
// does not fit (count_distinct needs the table for its lookups, records too wide for a
// tile / an LDS table) or a region outgrew its slack -- the caller merges through the table.
static const uint64_t kBucketedMergeMin = 1ull << 18;

static Status bucketed_merge(evql_query* q, evql_exchange* x, const MergeResolvedArgs& ma,
                             const uint64_t* identity, const uint64_t* d_recv,
                             const std::vector<uint64_t>& recv_rec,
                             const std::vector<uint64_t>& hbase, uint64_t total_rec, bool* done) {
  *done = false;
  hipStream_t s = q->ctx->stream;
  const uint32_t mw = ma.m.nwords, rw = mw + 1;
  if (total_rec < kBucketedMergeMin || recv_rec.size() > kMaxExchangeRanks) return Status();
  // LDS table of a bucket: <= 60 KB (30 KB tables = twice the buckets measured slower: 1.44 vs
  // 1.05 ms per 1e7 records), power of two slots + the two keyless groups
  uint32_t slots = 1;
  while (uint64_t(slots * 2 + 2) * (mw * 8 + 4) <= 60 * 1024) slots *= 2;
  if (slots < 64) return Status();
  uint32_t tile = uint32_t((56 * 1024) / (rw * 8)) / 256 * 256;
  if (tile > 2048) tile = 2048;
  if (tile < 256) return Status();
  int nsrc = 0;
  for (uint64_t c : recv_rec) nsrc += c ? 1 : 0;
  // buckets: a power of two with <= slots / 3 records on average, two levels of <= 256 bins
  uint32_t fbits = 1;
  while ((total_rec >> fbits) > slots / 3) ++fbits;
  const uint32_t c1_bits = (fbits + 1) / 2, c2_bits = fbits - c1_bits;
  if (c1_bits > 8) return Status();
  const uint32_t C1 = 1u << c1_bits, C2 = 1u << c2_bits;
  const uint64_t F = uint64_t(C1) * C2;
  // a group arrives once per source rank: the spread of a region is that of nsrc-fold copies
  auto region_cap = [&](double avg, double floor_) {
    return uint64_t(avg + 6.0 * std::sqrt(avg * std::max(nsrc, 1)) + floor_);
  };
  const uint64_t cap1 = region_cap(double(total_rec) / C1, 1024);
  const uint64_t cap2 = region_cap(double(total_rec) / double(F), 64);
  WsBuf<uint64_t> d_stage1(x, 14), d_stage2(x, 15), d_cur(x, 16);
  HIP_TRY(d_stage1.alloc(uint64_t(C1) * cap1 * rw * 8));
  HIP_TRY(d_stage2.alloc(F * cap2 * rw * 8));
  // cursors, then -- each on a line of its own -- the output counter (one add per bucket)
  // and the overflow flag (read by every workgroup of the later passes: next to the counter
  // those reads queued behind its atomics, 0.43 -> 2.7 ms for the merge pass)
  const uint64_t ncur = C1 + F + 64;
  HIP_TRY(d_cur.alloc(ncur * 8));
  HIP_TRY(hipMemsetAsync(d_cur, 0, ncur * 8, s));
  uint64_t* d_cnt1 = d_cur.p;
  uint64_t* d_cnt2 = d_cur.p + C1;
  uint64_t* d_out_count = d_cur.p + C1 + F + 16;
  uint32_t* d_flag = reinterpret_cast<uint32_t*>(d_cur.p + C1 + F + 48);
  if (q->mdense_cap < total_rec) {
    if (q->d_mdense) hipFree(q->d_mdense);
    q->d_mdense = nullptr;
    q->mdense_cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_mdense), total_rec * rw * 8));
    q->mdense_cap = total_rec;
  }
  BucketScatterArgs sa{};
  sa.in = d_recv;
  sa.out = d_stage1;
  sa.in_counts = nullptr;
  sa.n_in = total_rec;
  sa.in_region_cap = total_rec;
  sa.in_regions = 1;
  sa.tiles_per_region = uint32_t((total_rec + tile - 1) / tile);
  sa.tile = tile;
  sa.rw = rw;
  sa.rw_inv = uint32_t(((1ull << 32) + rw - 1) / rw);
  sa.shift = 64 - c1_bits;
  sa.bins = C1;
  sa.out_region_cap = cap1;
  sa.out_counts = d_cnt1;
  sa.status = d_flag;
  sa.first_value_word = 1 + ma.state_words;
  sa.ncols = ma.ncols;
  sa.str_mask = ma.str_mask;
  sa.nranks = uint32_t(recv_rec.size());
  uint64_t off = 0;
  for (size_t r = 0; r < recv_rec.size(); ++r) {
    sa.rank_start[r] = off;
    sa.heap_base[r] = hbase[r];
    off += recv_rec[r];
  }
  sa.rank_start[recv_rec.size()] = off;
  HIP_TRY(launch_bucket_scatter(sa, s));
  const bool two_levels = c2_bits > 0;
  BucketScatterArgs sb = sa;
  sb.in = d_stage1;
  sb.out = d_stage2;
  sb.in_counts = d_cnt1;
  sb.in_region_cap = cap1;
  sb.in_regions = C1;
  sb.tiles_per_region = uint32_t((cap1 + tile - 1) / tile);
  sb.shift = 64 - c1_bits - c2_bits;
  sb.bins = C2;
  sb.out_region_cap = cap2;
  sb.out_counts = d_cnt2;
  sb.nranks = 0;
  sb.str_mask = 0;
  if (two_levels) HIP_TRY(launch_bucket_scatter(sb, s));
  BucketMergeArgs ba{};
  ba.stage = two_levels ? d_stage2.p : d_stage1.p;
  ba.counts = two_levels ? d_cnt2 : d_cnt1;
  ba.region_cap = two_levels ? cap2 : cap1;
  ba.buckets = uint32_t(F);
  ba.mw = mw;
  ba.has_ident2 = ma.m.has_ident2;
  ba.state_words = ma.state_words;
  ba.first_row_word = ma.first_row_word;
  ba.ncols = ma.first_row_word == 0xffffffffu ? 0 : ma.ncols;
  ba.lds_slots = slots;
  for (uint32_t w = 0; w < mw; ++w) {
    ba.ops[w] = ma.m.ops[w];
    ba.identity[w] = identity[w];
  }
  ba.out = q->d_mdense;
  ba.out_cap = q->mdense_cap;
  ba.out_count = d_out_count;
  ba.status = d_flag;
  HIP_TRY(launch_bucket_merge(ba, s));
  uint64_t tail[33] = {0};  // [groups, ..., flag]
  HIP_TRY(hipMemcpyAsync(tail, d_out_count, sizeof(tail), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (tail[32] & 0xffffffffull) return Status();  // (a region / an LDS table overflowed)
  q->mdense_n = tail[0];
  q->merged_dense = true;
  x->stats.merge_buckets = F;
  *done = true;
  return Status();
}

// -----------------------------------------------------------------------------------------
// the exchange itself
// -----------------------------------------------------------------------------------------
Status exchange(evql_query* q, evql_exchange* x, int mode) {
  evql_ctx* ctx = q->ctx;
  evql_table* t = q->table;
  hipStream_t s = ctx->stream;
  const KernelPlan& kp = q->rplan();
  const int N = x->nranks;
  if (!q->executed) return Status::error(EVQL_EARG, "execute() was not called");
  if (N > int(kMaxExchangeRanks)) return Status::error(EVQL_EARG, "too many ranks");
  if (q->merged) return Status::error(EVQL_EARG, "the query was exchanged already");
  const uint32_t W = uint32_t(kp.words_per_slot());
  const uint32_t nc = uint32_t(kp.cols.size());
  const bool resolved = kp.need_first_row;
  const uint32_t rw_in = W + 1;
  const uint32_t rw = resolved ? rw_in + nc + 1 : rw_in;  // wire record words
  // (checked before anything is allocated or filled: the merged slot is W + nc + 1 words)
  if ((resolved ? W + nc + 1 : W) > uint32_t(kMaxStateWords + 3)) {
    return Status::error(EVQL_ENOTSUP, "too many words per merged group");
  }
  auto t0 = std::chrono::steady_clock::now();
  x->stats = evql_exchange_stats_t{};
  if (kp.n_exact > 0) {
    // EVQL_FLOAT_SUM_EXACT: the state words are integer multiples of 2^fsum_exp and are
    // added as they are -- every rank must have chosen the same quantum.  With
    // float_sum_bound = 0 each rank derives it from its OWN table's maxima, which may
    // fall on either side of a power of two.
    std::vector<uint64_t> mine_e(4), all_e(uint64_t(4) * N);
    for (int k = 0; k < 4; ++k) mine_e[k] = uint64_t(int64_t(q->fsum_exp[k]));
    int rce = x->tr.all_gather_u64(x->tr.user, mine_e.data(), 4, all_e.data());
    if (rce != EVQL_OK) return Status::error(rce, "exchange: all_gather of the sum quanta failed");
    for (int r = 0; r < N; ++r) {
      for (int k = 0; k < kp.n_exact; ++k) {
        if (all_e[uint64_t(r) * 4 + k] != mine_e[k]) {
          return Status::error(EVQL_EARG,
                               "exact float sums: the ranks chose different quanta (the tables' "
                               "maxima differ); pass the same float_sum_bound on every rank");
        }
      }
    }
  }

  // ---- 1. this rank's groups as dense records ---------------------------------------------
  const uint64_t n = q->ngroups;
  WsBuf<uint64_t> d_rec(x, 0);
  const uint64_t* d_records = nullptr;  // n records of rw_in words
  {
    RecordsView view;
    Status stv = query_records_view(q, &view);
    if (!stv.ok()) return stv;
    const uint64_t nd = view.nd;
    if (n && nd == n) {
      // (the partitioned path left every group as a dense record: read in place)
      d_records = view.dense;
    } else {
      HIP_TRY(d_rec.alloc(std::max<uint64_t>(n, 1) * rw_in * 8));
      if (nd) HIP_TRY(hipMemcpyAsync(d_rec, view.dense, nd * rw_in * 8, hipMemcpyDeviceToDevice, s));
      if (n > nd) {
        uint64_t* d_cnt = q->d_counters + 6;
        HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, s));
        HIP_TRY(launch_table_compact(q->d_gtab, q->gcap, q->gcap + 8, W, d_rec.p + nd * rw_in, n - nd,
                                     d_cnt, s));
      }
      d_records = d_rec;
    }
  }
  // ---- 2. first-row values into the records (plans that need them) -------------------------
  WsBuf<uint64_t> d_wire(x, 1);
  uint64_t str_mask = 0;
  std::vector<uint32_t> str_cols;
  if (resolved) {
    std::vector<RtColumn> rc(nc);
    for (uint32_t c = 0; c < nc; ++c) {
      const ColAccess& ca = kp.cols[c];
      rc[c] = RtColumn{};
      rc[c].pages = ca.layout_index >= 0 ? t->d_pages[ca.layout_index][0] : nullptr;
      rc[c].mode = ca.mode;
      rc[c].bits = ca.bits;
      if (q->nested) {
        rc[c].mode = ColAccess::SOA;  // (the 8-byte words stay beside a packed copy)
        rc[c].soa = ca.string_hash ? q->nested_strpos[c] : q->nested_flat[c];
      } else if (ca.packed) {
        const MaterializedColumn& m = t->materialized[ca.name];
        rc[c].pages = m.d_packed_pages;
        rc[c].base = m.d_packed;
      } else if (ca.mode == ColAccess::SOA) {
        const MaterializedColumn& m = t->materialized[ca.name];
        rc[c].soa = ca.string_hash ? m.d_strpos : m.d_values;
        rc[c].tags = m.d_tags;
      }
      if (ca.string_hash) {
        str_mask |= 1ull << c;
        str_cols.push_back(c);
      }
    }
    if (str_cols.size() > kMaxWireStrCols) {
      return Status::error(EVQL_ENOTSUP, "too many string columns in an exchanged plan");
    }
    WsBuf<RtColumn> d_cols(x, 2);
    HIP_TRY(d_cols.alloc(std::max<uint32_t>(nc, 1) * sizeof(RtColumn)));
    HIP_TRY(hipMemcpyAsync(d_cols, rc.data(), nc * sizeof(RtColumn), hipMemcpyHostToDevice, s));
    HIP_TRY(d_wire.alloc(std::max<uint64_t>(n, 1) * rw * 8));
    ResolveArgs ra{};
    ra.image = t->d_image;
    ra.cols = d_cols;
    ra.ncols = nc;
    ra.in_words = rw_in;
    ra.first_row_word = uint32_t(1 + kp.first_row_word());
    ra.rank_tag = uint64_t(x->rank) << 44;
    ra.in = d_records;
    ra.n = n;
    ra.out = d_wire;
    HIP_TRY(launch_resolve_records(ra, s));
    HIP_TRY(hipStreamSynchronize(s));  // (d_cols / rc live until here)
  }
  const uint64_t* d_src = resolved ? d_wire.p : d_records;

  // ---- 3. bucket by owner ---------------------------------------------------------------------
  std::vector<uint64_t> send_counts(N, 0), starts(N + 1, 0);
  WsBuf<uint64_t> d_send(x, 3), d_aux(x, 4);
  HIP_TRY(d_send.alloc(std::max<uint64_t>(n, 1) * rw * 8));
  const uint64_t* d_out = d_send.p;  // what is sent
  HIP_TRY(d_aux.alloc((3 * kMaxExchangeRanks + 4) * 8));
  uint64_t* d_counts = d_aux.p;
  uint64_t* d_starts = d_aux.p + kMaxExchangeRanks;
  uint64_t* d_cursors = d_aux.p + 2 * kMaxExchangeRanks + 2;
  if (mode == EVQL_EXCHANGE_BY_OWNER && N > 1) {
    HIP_TRY(hipMemsetAsync(d_aux, 0, (3 * kMaxExchangeRanks + 4) * 8, s));
    HIP_TRY(launch_owner_hist(d_src, n, rw, uint32_t(N), d_counts, s));
    HIP_TRY(hipMemcpyAsync(send_counts.data(), d_counts, N * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int r = 0; r < N; ++r) starts[r + 1] = starts[r] + send_counts[r];
    HIP_TRY(hipMemcpyAsync(d_starts, starts.data(), (N + 1) * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(launch_owner_scatter(d_src, n, rw, uint32_t(N), d_starts, d_cursors, d_send, s));
  } else {
    // every rank gets every record: "bucket" r = all of them (copied only when the string
    // words are about to be rewritten in place, step 4)
    if (resolved && !str_cols.empty()) {
      HIP_TRY(hipMemcpyAsync(d_send, d_src, n * rw * 8, hipMemcpyDeviceToDevice, s));
    } else {
      d_out = d_src;
    }
    for (int r = 0; r < N; ++r) send_counts[r] = n;
    starts[0] = 0;
    for (int r = 1; r <= N; ++r) starts[r] = n;  // (one segment)
  }
  const bool by_owner = mode == EVQL_EXCHANGE_BY_OWNER && N > 1;

  // ---- 4. string bytes of the records, in record order ------------------------------------------
  WsBuf<uint64_t> d_sizes(x, 5);
  WsBuf<uint8_t> d_heap(x, 6);
  std::vector<uint64_t> heap_counts(N, 0);  // bytes per destination (padded to words)
  uint64_t heap_bytes = 0;
  if (resolved && !str_cols.empty() && n) {
    WireStrArgs wa{};
    wa.image = t->d_image;
    wa.records = d_send;
    wa.n = n;
    wa.rw = rw;
    wa.tags_word = rw_in + nc;
    wa.nstr = uint32_t(str_cols.size());
    for (size_t k = 0; k < str_cols.size(); ++k) {
      wa.word[k] = rw_in + str_cols[k];
      wa.col[k] = str_cols[k];
      wa.pages[k] = t->d_pages[kp.cols[str_cols[k]].layout_index][0];
    }
    HIP_TRY(d_sizes.alloc((n + 2) * 8));
    wa.sizes = d_sizes;
    wa.nranks = by_owner ? uint32_t(N) : 1u;
    std::vector<uint64_t> seg(N + 1, 0);
    if (by_owner) {
      seg = starts;
    } else {
      seg[1] = n;
    }
    HIP_TRY(hipMemcpyAsync(d_starts, seg.data(), (N + 1) * 8, hipMemcpyHostToDevice, s));
    wa.starts = d_starts;
    HIP_TRY(launch_wire_str_sizes(wa, s));
    HIP_TRY(launch_exclusive_scan(d_sizes, n, d_sizes.p + n, s));
    HIP_TRY(hipMemcpyAsync(&heap_bytes, d_sizes.p + n, 8, hipMemcpyDeviceToHost, s));
    // segment borders in bytes
    std::vector<uint64_t> segoff(N + 1, 0);
    HIP_TRY(hipStreamSynchronize(s));
    const int nseg = by_owner ? N : 1;
    for (int r = 1; r <= nseg; ++r) {
      if (seg[r] >= n) {
        segoff[r] = heap_bytes;
      } else {
        HIP_TRY(hipMemcpy(&segoff[r], d_sizes.p + seg[r], 8, hipMemcpyDeviceToHost));
      }
    }
    HIP_TRY(d_heap.alloc(heap_bytes + 8 * (N + 1)));
    wa.heap = d_heap;
    HIP_TRY(launch_wire_str_copy(wa, s));
    for (int r = 0; r < N; ++r) {
      heap_counts[r] = by_owner ? segoff[r + 1] - segoff[r] : heap_bytes;
    }
    // (segments are sent as whole words: their starts have to be 8-aligned -- they are
    // re-packed below)
  }
  x->stats.export_ms = ms_since(t0);
  auto t1 = std::chrono::steady_clock::now();

  // ---- 5. counts, then the records (and the string bytes) --------------------------------------
  // per destination: [record count, heap bytes]
  std::vector<uint64_t> mine(2 * N), all(uint64_t(2) * N * N);
  for (int r = 0; r < N; ++r) {
    mine[2 * r] = by_owner ? send_counts[r] : n;
    mine[2 * r + 1] = heap_counts[r];
  }
  int rc = x->tr.all_gather_u64(x->tr.user, mine.data(), 2 * N, all.data());
  if (rc != EVQL_OK) return Status::error(rc, "exchange: all_gather of the counts failed");
  std::vector<uint64_t> recv_rec(N), recv_heap(N), send_words(N), recv_words(N);
  uint64_t total_rec = 0, total_heap_words = 0;
  for (int r = 0; r < N; ++r) {
    recv_rec[r] = all[uint64_t(r) * 2 * N + 2 * x->rank];
    recv_heap[r] = all[uint64_t(r) * 2 * N + 2 * x->rank + 1];
    total_rec += recv_rec[r];
    total_heap_words += (recv_heap[r] + 7) / 8;
    send_words[r] = mine[2 * r] * rw;
    recv_words[r] = recv_rec[r] * rw;
  }
  WsBuf<uint64_t> d_recv(x, 7);
  HIP_TRY(d_recv.alloc(std::max<uint64_t>(N == 1 ? 1 : total_rec, 1) * rw * 8));
  const uint64_t* d_in = d_recv.p;  // the received records, rank by rank
  if (by_owner) {
    rc = x->tr.all_to_all_words(x->tr.user, d_send, send_words.data(), d_recv, recv_words.data(), s);
  } else {
    // GATHER_ALL: the same n records to everybody -- as an all-to-all whose send
    // segments coincide (the transports read `send_counts[r]` words at the running
    // offset, so the buffer is replicated per destination only logically)
    if (N == 1) {
      // a single rank receives exactly what it sends: merged where it lies
      d_in = d_out;
    } else {
      WsBuf<uint64_t> d_rep(x, 8);
      HIP_TRY(d_rep.alloc(std::max<uint64_t>(n * rw * N, 1) * 8));
      for (int r = 0; r < N; ++r) {
        HIP_TRY(hipMemcpyAsync(d_rep.p + uint64_t(r) * n * rw, d_out, n * rw * 8,
                               hipMemcpyDeviceToDevice, s));
      }
      rc = x->tr.all_to_all_words(x->tr.user, d_rep, send_words.data(), d_recv, recv_words.data(), s);
    }
    if (rc == EVQL_OK) HIP_TRY(hipStreamSynchronize(s));
  }
  if (rc != EVQL_OK) return Status::error(rc, "exchange: transfer of the records failed");
  // string heaps: word-aligned segments
  WsBuf<uint64_t> d_hsend(x, 9), d_hrecv(x, 10);
  std::vector<uint64_t> hbase(N, 0);
  if (resolved && !str_cols.empty()) {
    std::vector<uint64_t> hs(N), hr(N);
    uint64_t tot = 0;
    for (int r = 0; r < N; ++r) {
      hs[r] = (heap_counts[r] + 7) / 8;
      hr[r] = (recv_heap[r] + 7) / 8;
      tot += hs[r];
    }
    HIP_TRY(d_hsend.alloc(std::max<uint64_t>(tot, 1) * 8));
    HIP_TRY(d_hrecv.alloc(std::max<uint64_t>(total_heap_words, 1) * 8));
    uint64_t woff = 0, boff = 0;
    for (int r = 0; r < N; ++r) {
      if (heap_counts[r]) {
        HIP_TRY(hipMemcpyAsync(d_hsend.p + woff, d_heap.p + (by_owner ? boff : 0), heap_counts[r],
                               hipMemcpyDeviceToDevice, s));
      }
      woff += hs[r];
      if (by_owner) boff += heap_counts[r];
    }
    rc = x->tr.all_to_all_words(x->tr.user, d_hsend, hs.data(), d_hrecv, hr.data(), s);
    if (rc != EVQL_OK) return Status::error(rc, "exchange: transfer of the strings failed");
    uint64_t b = 0;
    for (int r = 0; r < N; ++r) {
      hbase[r] = b;
      b += hr[r] * 8;
    }
  }
  HIP_TRY(hipStreamSynchronize(s));
  x->stats.transfer_ms = ms_since(t1);
  auto t2 = std::chrono::steady_clock::now();
  x->stats.groups_sent = by_owner ? n : n * uint64_t(N);
  x->stats.groups_received = total_rec;
  for (int r = 0; r < N; ++r) {
    if (r != x->rank) x->stats.bytes_sent += send_words[r] * 8 + heap_counts[r];
  }

  // ---- 6. merge, one batch per source rank, in rank order, into a fresh table --------------------
  const uint32_t mw = resolved ? W + nc + 1 : W;  // slot words of the merged table
  TableInitArgs ia{};
  ia.nwords = mw;
  ia.identity[0] = 0xFFFFFFFFFFFFFFFFull;
  int w = 1;
  if (kp.has_ident2()) ia.identity[w++] = 0xFFFFFFFFFFFFFFFFull;
  if (kp.need_first_row) ia.identity[w++] = 0xFFFFFFFFFFFFFFFFull;
  static const uint64_t kIdent[8] = {0, 0, 0xFFFFFFFFFFFFFFFFull, 0, 0x7FFFFFFFFFFFFFFFull,
                                     0x8000000000000000ull, 0x7FF0000000000000ull,
                                     0xFFF0000000000000ull};
  for (const auto& sw : kp.states) ia.identity[w++] = kIdent[sw.op & 7];
  for (; w < int(mw); ++w) ia.identity[w] = 0;
  MergeResolvedArgs ma{};
  ma.m.nwords = mw;
  ma.m.has_ident2 = kp.has_ident2() ? 1 : 0;
  w = 1;
  if (kp.has_ident2()) ma.m.ops[w++] = 255;
  if (kp.need_first_row) ma.m.ops[w++] = 2;
  for (const auto& sw : kp.states) ma.m.ops[w++] = uint32_t(sw.op);
  // a count_distinct word counts the pairs of the MERGED set: foreign counts are not
  // added, the pairs are exchanged below and counted again
  bool has_distinct = false;
  for (const auto& ag : kp.aggs) {
    if (ag.distinct_index >= 0) {
      ma.m.ops[kp.state_word_base() + ag.first_word] = kMergeSkip;
      has_distinct = true;
    }
  }
  ma.m.status = q->d_status;
  ma.state_words = W;
  ma.first_row_word = resolved ? uint32_t(kp.first_row_word()) : 0xffffffffu;
  ma.ncols = nc;
  ma.str_mask = str_mask;
  HIP_TRY(hipMemsetAsync(q->d_status, 0, 16, s));
  q->merged_dense = false;
  uint64_t ng = 0;
  uint64_t cap = 0;
  bool bucketed = false;
  if (!has_distinct) {
    Status stb = bucketed_merge(q, x, ma, ia.identity, d_in, recv_rec, hbase, total_rec, &bucketed);
    if (!stb.ok()) return stb;
  }
  if (bucketed) {
    ng = q->mdense_n;
    q->m_words = mw;
  } else {
    cap = 1 << 16;
    while (cap < total_rec * 2) cap <<= 1;
    if (!q->d_mtab || q->mcap != cap || q->m_words != mw) {
      if (q->d_mtab) hipFree(q->d_mtab);
      q->d_mtab = nullptr;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q->d_mtab), (cap + 8) * uint64_t(mw) * 8));
    }
    q->mcap = cap;
    q->m_words = mw;
    ia.words = q->d_mtab;
    ia.stride = cap + 8;
    HIP_TRY(launch_table_init(ia, s));
    ma.m.words = q->d_mtab;
    ma.m.gcap = cap;
    ma.m.stride = cap + 8;
    // the records of one source rank are its groups: pairwise different identities, one
    // launch per rank -- plain read-modify-write behind the identity CAS, new slots counted
    // as they are claimed
    uint64_t* d_ngroups = q->d_counters + 4;
    ma.m.exclusive = 1;
    ma.m.fresh = d_ngroups;
    HIP_TRY(hipMemsetAsync(d_ngroups, 0, 8, s));
    uint64_t roff = 0;
    for (int r = 0; r < N; ++r) {
      if (recv_rec[r]) {
        if (resolved) {
          ma.heap_base = hbase[r];
          HIP_TRY(launch_table_merge_resolved(ma, d_in + roff * rw, recv_rec[r], s));
        } else {
          HIP_TRY(launch_table_merge(ma.m, d_in + roff * rw, recv_rec[r], s));
        }
      }
      roff += recv_rec[r];
    }
    HIP_TRY(hipMemcpyAsync(&ng, d_ngroups, 8, hipMemcpyDeviceToHost, s));
  }
  // ---- 7. count_distinct: the pair sets follow their groups --------------------------------------
  for (const auto& ag : kp.aggs) {
    if (ag.distinct_index < 0) continue;
    Status st = exchange_pairset(q, x, by_owner, ag.distinct_index,
                                 uint32_t(kp.state_word_base() + ag.first_word), cap, mw);
    if (!st.ok()) return st;
  }
  uint32_t status[4] = {0};
  HIP_TRY(hipMemcpyAsync(status, q->d_status, 16, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (status[0] & 2u) return Status::error(EVQL_ENOMEM, "merged group table full");
  if (status[0] & 8u) return Status::error(EVQL_ENOMEM, "merged count_distinct set full");
  // the received string bytes are what the merged groups' string words point into
  q->m_heap.clear();
  if (resolved && !str_cols.empty() && total_heap_words) {
    q->m_heap.resize(total_heap_words * 8);
    HIP_TRY(hipMemcpy(q->m_heap.data(), d_hrecv, total_heap_words * 8, hipMemcpyDeviceToHost));
  }
  q->merged = true;
  q->ngroups = ng;
  q->stats.num_groups = ng;
  q->fetched = false;
  q->emit_pos = 0;
  x->stats.merge_ms = ms_since(t2);
  return Status();
}

}  // namespace

// -----------------------------------------------------------------------------------------
// chains: the tables of one partition scanned by one operator each (evql_query_create_chain)
// -----------------------------------------------------------------------------------------
// PartitionCursor hands the batches of its scans to ONE GroupByExpression, newest table
// first (server/sql/partition_cursor.cc:56-81, groupby.cc:69-185).  Here every table was
// aggregated by its own launch; their groups are merged like the partial aggregates of
// several ranks -- table i plays rank i: records in chain order into a fresh table, first
// rows resolved inside the table that produced them, (table << 44 | row) as the scan
// position, string bytes into one heap, count_distinct pairs re-inserted into one set.
namespace evql {

Status chain_merge(evql_query* head) {
  evql_ctx* ctx = head->ctx;
  hipStream_t s = ctx->stream;
  const KernelPlan& kp = head->rplan();
  std::vector<evql_query*> parts;
  parts.push_back(head);
  for (evql_query* c : head->chain) parts.push_back(c);
  const uint32_t W = uint32_t(kp.words_per_slot());
  const uint32_t nc = uint32_t(kp.cols.size());
  const bool resolved = kp.need_first_row;
  const uint32_t rw_in = W + 1;
  const uint32_t rw = resolved ? rw_in + nc + 1 : rw_in;
  const uint32_t mw = resolved ? W + nc + 1 : W;
  if (mw > uint32_t(kMaxStateWords + 3)) {
    return Status::error(EVQL_ENOTSUP, "too many words per merged group");
  }
  if (parts.size() >= (1u << 19)) return Status::error(EVQL_EARG, "too many tables in a chain");
  uint64_t total = 0, max_n = 0;
  for (evql_query* p : parts) {
    // (same plan everywhere: the slot layouts agree; a different KEY mode or word count
    // would mean different plans)
    // (a table whose string key has no usable dictionary keeps the hashed plan; its
    // records have the same rplan() layout as the coded tables')
    if (uint32_t(p->rplan().words_per_slot()) != W || p->rplan().cols.size() != nc ||
        p->rplan().key_mode != kp.key_mode || p->rplan().need_first_row != kp.need_first_row) {
      return Status::error(EVQL_ERUNTIME, "chain: the tables' plans disagree");
    }
    if (!p->executed) return Status::error(EVQL_EARG, "execute() was not called");
    total += p->ngroups;
    max_n = std::max(max_n, p->ngroups);
  }
  // ---- the merged table ---------------------------------------------------------------------
  uint64_t cap = 1 << 16;
  while (cap < total * 2) cap <<= 1;
  if (!head->d_mtab || head->mcap != cap || head->m_words != mw) {
    if (head->d_mtab) hipFree(head->d_mtab);
    head->d_mtab = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&head->d_mtab), (cap + 8) * uint64_t(mw) * 8));
  }
  head->mcap = cap;
  head->m_words = mw;
  TableInitArgs ia{};
  ia.words = head->d_mtab;
  ia.stride = cap + 8;
  ia.nwords = mw;
  ia.identity[0] = 0xFFFFFFFFFFFFFFFFull;
  int w = 1;
  if (kp.has_ident2()) ia.identity[w++] = 0xFFFFFFFFFFFFFFFFull;
  if (kp.need_first_row) ia.identity[w++] = 0xFFFFFFFFFFFFFFFFull;
  static const uint64_t kIdent[8] = {0, 0, 0xFFFFFFFFFFFFFFFFull, 0, 0x7FFFFFFFFFFFFFFFull,
                                     0x8000000000000000ull, 0x7FF0000000000000ull,
                                     0xFFF0000000000000ull};
  for (const auto& sw : kp.states) ia.identity[w++] = kIdent[sw.op & 7];
  for (; w < int(mw); ++w) ia.identity[w] = 0;
  HIP_TRY(launch_table_init(ia, s));
  MergeResolvedArgs ma{};
  ma.m.words = head->d_mtab;
  ma.m.gcap = cap;
  ma.m.stride = cap + 8;
  ma.m.nwords = mw;
  ma.m.has_ident2 = kp.has_ident2() ? 1 : 0;
  w = 1;
  if (kp.has_ident2()) ma.m.ops[w++] = 255;
  if (kp.need_first_row) ma.m.ops[w++] = 2;
  for (const auto& sw : kp.states) ma.m.ops[w++] = uint32_t(sw.op);
  for (const auto& ag : kp.aggs) {
    if (ag.distinct_index >= 0) ma.m.ops[kp.state_word_base() + ag.first_word] = kMergeSkip;
  }
  ma.m.status = head->d_status;
  ma.state_words = W;
  ma.first_row_word = resolved ? uint32_t(kp.first_row_word()) : 0xffffffffu;
  ma.ncols = nc;
  uint64_t* d_ngroups = head->d_counters + 4;
  ma.m.exclusive = 1;  // (one launch per table, a table's groups are pairwise different)
  ma.m.fresh = d_ngroups;
  HIP_TRY(hipMemsetAsync(d_ngroups, 0, 8, s));
  HIP_TRY(hipMemsetAsync(head->d_status, 0, 16, s));

  // ---- table by table: records -> wire form -> merge -----------------------------------------
  DevBuf<uint64_t> d_rec, d_wire, d_sizes;
  DevBuf<RtColumn> d_cols;
  DevBuf<uint8_t> d_heap;
  HIP_TRY(d_rec.alloc(std::max<uint64_t>(max_n, 1) * rw_in * 8));
  if (resolved) {
    HIP_TRY(d_wire.alloc(std::max<uint64_t>(max_n, 1) * rw * 8));
    HIP_TRY(d_cols.alloc(std::max<uint32_t>(nc, 1) * sizeof(RtColumn)));
  }
  head->m_heap.clear();
  uint64_t rows_scanned = 0, rows_passed = 0;
  double kernel_ms = 0;
  for (size_t pi = 0; pi < parts.size(); ++pi) {
    evql_query* q = parts[pi];
    evql_table* t = q->table;
    rows_scanned += q->stats.rows_scanned;
    rows_passed += q->stats.rows_passed;
    kernel_ms += q->stats.kernel_ms;
    const uint64_t n = q->ngroups;
    if (n == 0) continue;
    RecordsView view;
    Status stv = query_records_view(q, &view);
    if (!stv.ok()) return stv;
    const uint64_t nd = view.nd;
    if (nd) HIP_TRY(hipMemcpyAsync(d_rec, view.dense, nd * rw_in * 8, hipMemcpyDeviceToDevice, s));
    if (n > nd) {
      uint64_t* d_cnt = q->d_counters + 6;
      HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, s));
      HIP_TRY(launch_table_compact(q->d_gtab, q->gcap, q->gcap + 8, W, d_rec.p + nd * rw_in, n - nd,
                                   d_cnt, s));
    }
    if (!resolved) {
      HIP_TRY(launch_table_merge(ma.m, d_rec, n, s));
      continue;
    }
    std::vector<RtColumn> rc(nc);
    std::vector<uint32_t> str_cols;
    uint64_t str_mask = 0;
    for (uint32_t c = 0; c < nc; ++c) {
      const ColAccess& ca = q->rplan().cols[c];
      rc[c] = RtColumn{};
      rc[c].pages = ca.layout_index >= 0 ? t->d_pages[ca.layout_index][0] : nullptr;
      rc[c].mode = ca.mode;
      rc[c].bits = ca.bits;
      if (ca.packed) {
        const MaterializedColumn& m = t->materialized[ca.name];
        rc[c].pages = m.d_packed_pages;
        rc[c].base = m.d_packed;
      } else if (ca.mode == ColAccess::SOA) {
        const MaterializedColumn& m = t->materialized[ca.name];
        rc[c].soa = ca.string_hash ? m.d_strpos : m.d_values;
        rc[c].tags = m.d_tags;
      }
      if (ca.string_hash) {
        str_mask |= 1ull << c;
        str_cols.push_back(c);
      }
    }
    if (str_cols.size() > kMaxWireStrCols) {
      return Status::error(EVQL_ENOTSUP, "too many string columns in a chain plan");
    }
    HIP_TRY(hipMemcpyAsync(d_cols, rc.data(), nc * sizeof(RtColumn), hipMemcpyHostToDevice, s));
    ResolveArgs ra{};
    ra.image = t->d_image;
    ra.cols = d_cols;
    ra.ncols = nc;
    ra.in_words = rw_in;
    ra.first_row_word = uint32_t(1 + kp.first_row_word());
    ra.rank_tag = uint64_t(pi) << 44;
    ra.in = d_rec;
    ra.n = n;
    ra.out = d_wire;
    HIP_TRY(launch_resolve_records(ra, s));
    uint64_t heap_bytes = 0;
    if (!str_cols.empty()) {
      WireStrArgs wa{};
      wa.image = t->d_image;
      wa.records = d_wire;
      wa.n = n;
      wa.rw = rw;
      wa.tags_word = rw_in + nc;
      wa.nstr = uint32_t(str_cols.size());
      for (size_t k = 0; k < str_cols.size(); ++k) {
        wa.word[k] = rw_in + str_cols[k];
        wa.col[k] = str_cols[k];
        wa.pages[k] = t->d_pages[q->rplan().cols[str_cols[k]].layout_index][0];
      }
      HIP_TRY(d_sizes.alloc((n + 4) * 8));
      wa.sizes = d_sizes;
      wa.nranks = 1;
      const uint64_t seg[2] = {0, n};
      uint64_t* d_seg = d_sizes.p + n + 2;
      HIP_TRY(hipMemcpyAsync(d_seg, seg, 16, hipMemcpyHostToDevice, s));
      wa.starts = d_seg;
      HIP_TRY(launch_wire_str_sizes(wa, s));
      HIP_TRY(launch_exclusive_scan(d_sizes, n, d_sizes.p + n, s));
      HIP_TRY(hipMemcpyAsync(&heap_bytes, d_sizes.p + n, 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      HIP_TRY(d_heap.alloc(heap_bytes + 16));
      wa.heap = d_heap;
      HIP_TRY(launch_wire_str_copy(wa, s));
    }
    ma.str_mask = str_mask;
    ma.heap_base = head->m_heap.size();
    HIP_TRY(launch_table_merge_resolved(ma, d_wire, n, s));
    HIP_TRY(hipStreamSynchronize(s));  // (rc / seg live until here)
    if (heap_bytes) {
      const size_t old = head->m_heap.size();
      head->m_heap.resize(old + heap_bytes);
      HIP_TRY(hipMemcpy(head->m_heap.data() + old, d_heap, heap_bytes, hipMemcpyDeviceToHost));
    }
  }
  if (head->m_heap.size() > kStrOffMask) {
    return Status::error(EVQL_ENOTSUP, "chain: first-row strings exceed the offset range");
  }
  // ---- count_distinct: the union of the tables' pair sets ---------------------------------------
  if (kp.n_distinct > 0) {
    std::vector<uint64_t> counts(size_t(kp.n_distinct) * parts.size(), 0);
    uint64_t max_total = 0;
    for (int d = 0; d < kp.n_distinct; ++d) {
      uint64_t tot = 0;
      for (size_t pi = 0; pi < parts.size(); ++pi) {
        evql_query* q = parts[pi];
        if (!q->d_pairset[d]) continue;
        uint64_t* d_cnt = q->d_counters + 6;
        HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, s));
        HIP_TRY(launch_pairset_export(q->d_pairset[d], q->pairset_cap, nullptr, 0, d_cnt, s));
        HIP_TRY(hipMemcpyAsync(&counts[size_t(d) * parts.size() + pi], d_cnt, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        tot += counts[size_t(d) * parts.size() + pi];
      }
      max_total = std::max(max_total, tot);
    }
    uint64_t set_cap = 1 << 16;
    while (set_cap < max_total * 2) set_cap <<= 1;
    for (const auto& ag : kp.aggs) {
      if (ag.distinct_index < 0) continue;
      const int d = ag.distinct_index;
      uint64_t tot = 0;
      for (size_t pi = 0; pi < parts.size(); ++pi) tot += counts[size_t(d) * parts.size() + pi];
      DevBuf<uint64_t> d_tr;
      HIP_TRY(d_tr.alloc(std::max<uint64_t>(tot, 1) * 24));
      uint64_t off = 0;
      for (size_t pi = 0; pi < parts.size(); ++pi) {
        evql_query* q = parts[pi];
        const uint64_t np = counts[size_t(d) * parts.size() + pi];
        if (!np) continue;
        uint64_t* d_cnt = q->d_counters + 6;
        HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, s));
        HIP_TRY(launch_pairset_export(q->d_pairset[d], q->pairset_cap, d_tr.p + off * 3, np, d_cnt, s));
        off += np;
      }
      if (head->d_mset[d] && head->mset_cap[d] != set_cap) {
        hipFree(head->d_mset[d]);
        head->d_mset[d] = nullptr;
      }
      if (!head->d_mset[d]) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&head->d_mset[d]), set_cap * 24));
      }
      head->mset_cap[d] = set_cap;
      uint64_t* d_set = head->d_mset[d];
      HIP_TRY(hipMemsetAsync(d_set, 0xff, set_cap * 24, s));
      PairsetMergeArgs pa{};
      pa.set = d_set;
      pa.set_cap = set_cap;
      pa.words = head->d_mtab;
      pa.gcap = cap;
      pa.nwords = mw;
      pa.word = uint32_t(kp.state_word_base() + ag.first_word);
      pa.key_mode = uint32_t(kp.key_mode);
      pa.status = head->d_status;
      HIP_TRY(launch_pairset_merge(pa, d_tr, tot, s));
      HIP_TRY(hipStreamSynchronize(s));
      // (PARTIAL emission reads the values of every group from the merged set,
      // fetch_results)
    }
  }
  uint32_t status[4] = {0};
  HIP_TRY(hipMemcpyAsync(status, head->d_status, 16, hipMemcpyDeviceToHost, s));
  uint64_t ng = 0;
  HIP_TRY(hipMemcpyAsync(&ng, d_ngroups, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (status[0] & 2u) return Status::error(EVQL_ENOMEM, "merged group table full");
  if (status[0] & 8u) return Status::error(EVQL_ENOMEM, "merged count_distinct set full");
  head->merged = true;
  head->ngroups = ng;
  head->stats.num_groups = ng;
  head->stats.rows_scanned = rows_scanned;
  head->stats.rows_passed = rows_passed;
  head->stats.kernel_ms = kernel_ms;
  head->stats.total_ms = kernel_ms;
  head->fetched = false;
  head->emit_pos = 0;
  return Status();
}

}  // namespace evql

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

int evql_exchange_create(evql_ctx_t* ctx, int nranks, int rank, const evql_transport_t* transport,
                         evql_exchange_t** out) {
  if (!ctx || !transport || !out || nranks < 1 || rank < 0 || rank >= nranks ||
      !transport->all_gather_u64 || !transport->all_to_all_words) {
    return fail(EVQL_EARG, "bad arguments");
  }
  evql_exchange* x = new evql_exchange();
  x->ctx = ctx;
  x->nranks = nranks;
  x->rank = rank;
  x->tr = *transport;
  x->name = transport->name ? transport->name : "custom";
  *out = x;
  return EVQL_OK;
}

int evql_hub_create(int nranks, evql_hub_t** out) {
  if (nranks < 1 || !out) return fail(EVQL_EARG, "bad arguments");
  evql_hub* h = new evql_hub();
  h->nranks = nranks;
  h->host_send.assign(nranks, nullptr);
  h->dev_send.assign(nranks, nullptr);
  h->send_counts.assign(nranks, std::vector<uint64_t>(nranks, 0));
  *out = h;
  return EVQL_OK;
}

void evql_hub_destroy(evql_hub_t* hub) { delete hub; }

int evql_exchange_create_hub(evql_ctx_t* ctx, evql_hub_t* hub, int rank, evql_exchange_t** out) {
  if (!ctx || !hub || !out || rank < 0 || rank >= hub->nranks) return fail(EVQL_EARG, "bad arguments");
  evql_exchange* x = new evql_exchange();
  x->ctx = ctx;
  x->nranks = hub->nranks;
  x->rank = rank;
  x->hub = hub;
  x->tr.user = x;
  x->tr.all_gather_u64 = hub_all_gather;
  x->tr.all_to_all_words = hub_all_to_all;
  x->name = "hub";
  *out = x;
  return EVQL_OK;
}

int evql_rccl_unique_id(void* id128) {
  if (!id128) return fail(EVQL_EARG, "null argument");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return fail(EVQL_EDEVICE, "ncclGetUniqueId failed");
  memcpy(id128, &id, sizeof(id));
  return EVQL_OK;
}

int evql_exchange_create_rccl(evql_ctx_t* ctx, int nranks, int rank, const void* id128,
                              evql_exchange_t** out) {
  if (!ctx || !id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) {
    return fail(EVQL_EARG, "bad arguments");
  }
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(EVQL_EDEVICE, "hipSetDevice failed");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  evql_exchange* x = new evql_exchange();
  x->ctx = ctx;
  x->nranks = nranks;
  x->rank = rank;
  if (ncclCommInitRank(&x->comm, nranks, id, rank) != ncclSuccess) {
    delete x;
    return fail(EVQL_EDEVICE, "ncclCommInitRank failed");
  }
  if (hipMalloc(reinterpret_cast<void**>(&x->d_scratch), (512 + 512 * kMaxExchangeRanks) * 8) !=
      hipSuccess) {
    ncclCommDestroy(x->comm);
    delete x;
    return fail(EVQL_EDEVICE, "hipMalloc failed");
  }
  x->tr.user = x;
  x->tr.all_gather_u64 = rccl_all_gather;
  x->tr.all_to_all_words = rccl_all_to_all;
  x->name = "rccl";
  *out = x;
  return EVQL_OK;
}

void evql_exchange_destroy(evql_exchange_t* x) {
  if (!x) return;
  if (x->comm) ncclCommDestroy(x->comm);
  if (x->d_scratch) hipFree(x->d_scratch);
  delete x;
}

const char* evql_exchange_backend(const evql_exchange_t* x) { return x ? x->name.c_str() : ""; }

int evql_query_exchange(evql_query_t* q, evql_exchange_t* x, int mode) {
  if (!q || !x) return fail(EVQL_EARG, "null argument");
  if (mode != EVQL_EXCHANGE_GATHER_ALL && mode != EVQL_EXCHANGE_BY_OWNER) {
    return fail(EVQL_EARG, "bad exchange mode");
  }
  try {
    if (hipSetDevice(q->ctx->device) != hipSuccess) return fail(EVQL_EDEVICE, "hipSetDevice failed");
    Status st = exchange(q, x, mode);
    if (!st.ok()) return fail(st.code, st.msg);
    return EVQL_OK;
  } catch (const std::exception& e) {
    return fail(EVQL_ERUNTIME, e.what());
  }
}

int evql_exchange_last_stats(const evql_exchange_t* x, evql_exchange_stats_t* out) {
  if (!x || !out) return fail(EVQL_EARG, "null argument");
  *out = x->stats;
  return EVQL_OK;
}

}  // extern "C"
