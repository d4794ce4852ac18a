// evql_host.hpp -- host-side C++ mirror of the reference's operator interface
// for the hot path, built on nothing but the C ABI (evql_gpu.h).
//
// The reference adapter derives from csql::TableExpression and
// forwards to the same five C calls; this header carries a *mirror* of that tiny
// interface so the operator can be exercised without the reference tree (which
// does not exist on the GPU box):
//
//   csql::SType            sql/svalue.h:41-49
//   csql::SVector          sql/svalue.h:163-201, sql/svalue.cc:410-517
//   ReturnCode             util/return_code.h:32-80
//   csql::TableExpression  sql/table_expression.h:35-50
//   kOutputBatchSize       sql/CSTableScan.h:46, statements/select/groupby.h:36
//   ResultCursor           sql/result_cursor.cc:34-100 (execute once, pull batches)
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>
#include "evql_gpu.h"

namespace evql_host {

enum class SType : uint8_t { NIL, UINT64, INT64, FLOAT64, BOOL, STRING, TIMESTAMP64 };

static const size_t kOutputBatchSize = 1024;

class ReturnCode {
 public:
  static ReturnCode success() { return ReturnCode(true, "", ""); }
  static ReturnCode error(const std::string& code, const std::string& msg) {
    return ReturnCode(false, code, msg);
  }
  bool isSuccess() const { return ok_; }
  const std::string& getCode() const { return code_; }
  const std::string& getMessage() const { return msg_; }

 private:
  ReturnCode(bool ok, const std::string& c, const std::string& m) : ok_(ok), code_(c), msg_(m) {}
  bool ok_;
  std::string code_, msg_;
};

// packed column batch: elements back to back, value bytes + 1 tag byte
class SVector {
 public:
  explicit SVector(SType type) : type_(type), data_(nullptr), capacity_(0), size_(0) {}
  SVector(const SVector&) = delete;
  SVector& operator=(const SVector&) = delete;
  SVector(SVector&& o) : type_(o.type_), data_(o.data_), capacity_(o.capacity_), size_(o.size_) {
    o.data_ = nullptr;
    o.capacity_ = o.size_ = 0;
  }
  ~SVector() { free(data_); }
  SType getType() const { return type_; }
  const void* getData() const { return data_; }
  size_t getSize() const { return size_; }
  size_t getCapacity() const { return capacity_; }
  void clear() { size_ = 0; }
  void increaseCapacity(size_t min_capacity) {  // exact realloc, svalue.cc:463-476
    if (min_capacity <= capacity_) return;
    data_ = realloc(data_, min_capacity);
    if (!data_) throw std::bad_alloc();
    capacity_ = min_capacity;
  }
  void append(const void* data, size_t size) {
    if (size_ + size > capacity_) increaseCapacity(size_ + size);
    memcpy(static_cast<char*>(data_) + size_, data, size);
    size_ += size;
  }

 private:
  SType type_;
  void* data_;
  size_t capacity_;
  size_t size_;
};

class TableExpression {
 public:
  virtual ~TableExpression() = default;
  virtual ReturnCode execute() = 0;
  virtual ReturnCode nextBatch(SVector* columns, size_t* len) = 0;
  virtual size_t getColumnCount() const = 0;
  virtual SType getColumnType(size_t idx) const = 0;
};

inline const char* status_code_string(int rc) {
  switch (rc) {
    case EVQL_EIO: return "EIO";
    case EVQL_EARG: return "EARG";
    case EVQL_ENOTSUP: return "ENOTSUP";
    case EVQL_EDEVICE: return "EDEVICE";
    case EVQL_ENOMEM: return "ENOMEM";
    default: return "ERUNTIME";
  }
}

// The fused operator: GroupByExpression over FastCSTableScan, on the MI355X.
// Construction lowers + compiles the plan; a plan that cannot be lowered throws
// NotLowerable so that the scheduler override can build the CPU operators
// instead (DefaultScheduler::buildGroupByExpression fallback).
class NotLowerable : public std::runtime_error {
 public:
  using std::runtime_error::runtime_error;
};

class GpuGroupByScan : public TableExpression {
 public:
  using Heartbeat = std::function<ReturnCode()>;  // Transaction::triggerHeartbeat

  GpuGroupByScan(evql_ctx_t* ctx, evql_table_t* table, const evql_plan_desc_t& plan,
                 Heartbeat heartbeat = nullptr)
      : query_(nullptr), heartbeat_(std::move(heartbeat)) {
    int rc = evql_query_create(ctx, table, &plan, &query_);
    if (rc == EVQL_ENOTSUP) throw NotLowerable(evql_last_error());
    if (rc != EVQL_OK) throw std::runtime_error(evql_last_error());
    const int n = evql_query_column_count(query_);
    for (int i = 0; i < n; ++i) types_.push_back(SType(evql_query_column_type(query_, i)));
    bufs_.resize(types_.size());
  }
  ~GpuGroupByScan() override { evql_query_destroy(query_); }
  GpuGroupByScan(const GpuGroupByScan&) = delete;
  GpuGroupByScan& operator=(const GpuGroupByScan&) = delete;

  ReturnCode execute() override {
    int rc = evql_query_execute(query_, heartbeat_ ? &GpuGroupByScan::heartbeat_thunk : nullptr,
                                this);
    if (rc != EVQL_OK) return ReturnCode::error(status_code_string(rc), evql_last_error());
    return ReturnCode::success();
  }

  // appends up to kOutputBatchSize rows; *len == 0 => EOF (and stays 0)
  ReturnCode nextBatch(SVector* columns, size_t* len) override {
    size_t n = 0;
    int rc = evql_query_next_batch(query_, kOutputBatchSize, bufs_.data(), &n);
    if (rc != EVQL_OK) return ReturnCode::error(status_code_string(rc), evql_last_error());
    for (size_t i = 0; i < types_.size(); ++i) {
      if (bufs_[i].size) columns[i].append(bufs_[i].data, bufs_[i].size);
    }
    *len = n;
    return ReturnCode::success();
  }

  size_t getColumnCount() const override { return types_.size(); }
  SType getColumnType(size_t idx) const override { return types_.at(idx); }

  evql_query_t* handle() { return query_; }

  // Fuses LimitExpression(limit, offset, OrderByExpression(specs, this)) into the
  // operator (orderby.cc:60-160, limit.cc:52-125); limit < 0 = ORDER BY only.
  // Throws NotLowerable when the first sort key cannot be read on the device, in
  // which case the scheduler stacks the CPU OrderBy / Limit operators on top.
  void setOrder(const std::vector<evql_sort_spec_t>& specs, int64_t limit, uint64_t offset) {
    int rc = evql_query_set_order(query_, specs.data(), uint32_t(specs.size()), limit, offset);
    if (rc == EVQL_ENOTSUP) throw NotLowerable(evql_last_error());
    if (rc != EVQL_OK) throw std::runtime_error(evql_last_error());
  }

 private:
  static int heartbeat_thunk(void* self) {
    auto* s = static_cast<GpuGroupByScan*>(self);
    return s->heartbeat_().isSuccess() ? 0 : 1;
  }
  evql_query_t* query_;
  Heartbeat heartbeat_;
  std::vector<SType> types_;
  std::vector<evql_column_buf_t> bufs_;
};

// GroupByMergeExpression (groupby.h:130-178, groupby.cc:493-672) for partial
// aggregates that arrive as bytes.  The reference pulls the frames itself from
// its RPC scheduler inside execute(); here the transport hands them over with
// addFrame() (or the two STRING SVectors of a partial operator with addPart())
// before execute().
class GroupByMerge : public TableExpression {
 public:
  explicit GroupByMerge(const evql_plan_desc_t& plan) : merge_(nullptr) {
    int rc = evql_merge_create(&plan, &merge_);
    if (rc == EVQL_ENOTSUP) throw NotLowerable(evql_last_error());
    if (rc != EVQL_OK) throw std::runtime_error(evql_last_error());
    const size_t n = evql_merge_column_count(merge_);
    for (size_t i = 0; i < n; ++i) types_.push_back(SType(evql_merge_column_type(merge_, i)));
    bufs_.resize(types_.size());
  }
  ~GroupByMerge() override { evql_merge_destroy(merge_); }
  GroupByMerge(const GroupByMerge&) = delete;
  GroupByMerge& operator=(const GroupByMerge&) = delete;

  // payload of one EVQL_OP_QUERY_PARTIALAGGR_RESULT frame
  ReturnCode addFrame(const void* payload, size_t len) {
    int rc = evql_merge_add_frame(merge_, payload, len);
    if (rc != EVQL_OK) return ReturnCode::error(status_code_string(rc), evql_last_error());
    return ReturnCode::success();
  }
  // drains a PartialGroupBy-shaped operator (2 STRING columns: key, data)
  ReturnCode addPart(TableExpression* partial) {
    ReturnCode rc = partial->execute();
    if (!rc.isSuccess()) return rc;
    for (;;) {
      SVector cols[2] = {SVector(SType::STRING), SVector(SType::STRING)};
      size_t n = 0;
      rc = partial->nextBatch(cols, &n);
      if (!rc.isSuccess()) return rc;
      if (n == 0) return ReturnCode::success();
      int r = evql_merge_add_rows(merge_, cols[0].getData(), cols[0].getSize(), cols[1].getData(),
                                  cols[1].getSize(), n);
      if (r != EVQL_OK) return ReturnCode::error(status_code_string(r), evql_last_error());
    }
  }

  ReturnCode execute() override { return ReturnCode::success(); }
  ReturnCode nextBatch(SVector* columns, size_t* len) override {
    size_t n = 0;
    int rc = evql_merge_next_batch(merge_, kOutputBatchSize, bufs_.data(), &n);
    if (rc != EVQL_OK) return ReturnCode::error(status_code_string(rc), evql_last_error());
    for (size_t i = 0; i < types_.size(); ++i) {
      if (bufs_[i].size) columns[i].append(bufs_[i].data, bufs_[i].size);
    }
    *len = n;
    return ReturnCode::success();
  }
  size_t getColumnCount() const override { return types_.size(); }
  SType getColumnType(size_t idx) const override { return types_.at(idx); }

 private:
  evql_merge_t* merge_;
  std::vector<SType> types_;
  std::vector<evql_column_buf_t> bufs_;
};

// pull cursor over any TableExpression (result_cursor.cc:34-100)
class ResultCursor {
 public:
  explicit ResultCursor(TableExpression* e) : expr_(e), pos_(0), len_(0), eof_(false) {
    for (size_t i = 0; i < e->getColumnCount(); ++i) cols_.emplace_back(e->getColumnType(i));
    ReturnCode rc = e->execute();
    if (!rc.isSuccess()) throw std::runtime_error(rc.getMessage());
  }
  // fetches the next batch; false at EOF
  bool nextBatch() {
    if (eof_) return false;
    for (auto& c : cols_) c.clear();
    ReturnCode rc = expr_->nextBatch(cols_.data(), &len_);
    if (!rc.isSuccess()) throw std::runtime_error(rc.getMessage());
    if (len_ == 0) eof_ = true;
    return !eof_;
  }
  size_t batchLength() const { return len_; }
  const SVector& column(size_t i) const { return cols_[i]; }

 private:
  TableExpression* expr_;
  std::vector<SVector> cols_;
  size_t pos_, len_;
  bool eof_;
};

}  // namespace evql_host
