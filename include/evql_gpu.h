/*
 * evql_gpu.h -- C ABI of the MI355X scan -> filter -> GROUP BY executor.
 *
 * This is the drop-in boundary: plain pointers and
 * sizes, no C++/torch types, never throws.  Each entry point names the
 * reference interface (file:line under 17ai/eventql) it stands in for; the
 * reference-side adapter (a csql::TableExpression subclass and a
 * csql::DefaultScheduler override) that binds these is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns EVQL_OK (0) or a negative evql_status; the message
 *     of the last failure on the calling thread is evql_last_error()
 *     (reference: ReturnCode::error(code,msg), util/return_code.h:32-80)
 *   - one thread drives one query object (reference threading model: pull
 *     operators are never called concurrently, SURVEY 8b "Threading")
 *   - there is NO CPU fallback behind this ABI: if the HIP device or the kernel
 *     compiler is unavailable every compute entry point fails with
 *     EVQL_EDEVICE.
 */
#ifndef EVQL_GPU_H
#define EVQL_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* status codes (string codes of util/return_code.h mapped to integers)      */
/* ------------------------------------------------------------------------ */
typedef enum {
  EVQL_OK = 0,
  EVQL_EIO = -1,      /* "EIO"      */
  EVQL_EARG = -2,     /* "EARG"     */
  EVQL_ERUNTIME = -3, /* "ERUNTIME" (incl. "division by zero") */
  EVQL_ENOTSUP = -4,  /* plan not lowerable: caller falls back to CPU operators */
  EVQL_EDEVICE = -5,  /* HIP device / hiprtc unavailable or failed */
  EVQL_ENOMEM = -6
} evql_status;

const char* evql_last_error(void);
const char* evql_version(void);

/* ------------------------------------------------------------------------ */
/* value model: mirrors csql::SType / STag (sql/svalue.h:41-56)               */
/* ------------------------------------------------------------------------ */
typedef enum {
  EVQL_T_NIL = 0,
  EVQL_T_UINT64 = 1,
  EVQL_T_INT64 = 2,
  EVQL_T_FLOAT64 = 3,
  EVQL_T_BOOL = 4,
  EVQL_T_STRING = 5,
  EVQL_T_TIMESTAMP64 = 6
} evql_stype;

#define EVQL_STAG_NULL 1

/* cstable column model (io/cstable/cstable.h:113-131) */
typedef enum {
  EVQL_COL_SUBRECORD = 0,
  EVQL_COL_BOOLEAN = 1,
  EVQL_COL_UNSIGNED_INT = 2,
  EVQL_COL_SIGNED_INT = 3,
  EVQL_COL_STRING = 4,
  EVQL_COL_FLOAT = 5,
  EVQL_COL_DATETIME = 6
} evql_column_type;

typedef enum {
  EVQL_ENC_BOOLEAN_BITPACKED = 1,
  EVQL_ENC_UINT32_BITPACKED = 10,
  EVQL_ENC_UINT32_PLAIN = 11,
  EVQL_ENC_UINT64_PLAIN = 12,
  EVQL_ENC_UINT64_LEB128 = 13,
  EVQL_ENC_FLOAT_IEEE754 = 14,
  EVQL_ENC_STRING_PLAIN = 100
} evql_column_encoding;

/* ------------------------------------------------------------------------ */
/* bytecode: mirrors csql::vm::Instruction / vm::Program (runtime/vm.h:44-82) */
/* ------------------------------------------------------------------------ */
typedef enum {
  EVQL_X_CALL_PURE = 1,
  EVQL_X_CALL_INSTANCE = 2,
  EVQL_X_LITERAL = 3,
  EVQL_X_INPUT = 4,
  EVQL_X_JUMP = 5,
  EVQL_X_CJUMP = 6,
  EVQL_X_RETURN = 7
} evql_opcode;

/*
 * Pure functions: the reference stores a C function pointer in
 * Instruction::arg0; across the ABI it becomes one of these ids.  The adapter
 * maps by symbol string "name#ret/arg;arg;" (runtime/symboltable.cc:33-41).
 * Numbering: family * 16 + type-slot.
 */
typedef enum {
  EVQL_TS_UINT64 = 0,
  EVQL_TS_INT64 = 1,
  EVQL_TS_FLOAT64 = 2,
  EVQL_TS_BOOL = 3,
  EVQL_TS_STRING = 4,
  EVQL_TS_TIMESTAMP64 = 5,
  EVQL_TS_NIL = 6
} evql_type_slot;

typedef enum {
  EVQL_FAM_LOGICAL_AND = 1, /* boolean.cc:38  */
  EVQL_FAM_LOGICAL_OR = 2,  /* boolean.cc:52  */
  EVQL_FAM_NEG = 3,         /* boolean.cc:66  */
  EVQL_FAM_CMP = 4,         /* boolean.cc:81-180  -> int64 -1/0/1 */
  EVQL_FAM_EQ = 5,
  EVQL_FAM_NEQ = 6,
  EVQL_FAM_LT = 7,
  EVQL_FAM_LTE = 8,
  EVQL_FAM_GT = 9,
  EVQL_FAM_GTE = 10,
  EVQL_FAM_ADD = 11, /* math.cc:34-  */
  EVQL_FAM_SUB = 12,
  EVQL_FAM_MUL = 13,
  EVQL_FAM_DIV = 14, /* int/uint division by zero => ERUNTIME */
  EVQL_FAM_MOD = 15,
  EVQL_FAM_POW = 16,
  EVQL_FAM_TO_NIL = 17,   /* conversion.cc:34-93   */
  EVQL_FAM_TO_INT64 = 18, /* conversion.cc:96-137  */
  EVQL_FAM_TO_TIMESTAMP64 = 19,
  /* string functions (expressions/string.cc, conversion.cc:140-215).  They PRODUCE (or
   * read) strings and are evaluated where the reference evaluates a GROUP BY's select
   * list: once per group, at emission (groupby.cc:187-220), on the host.  In WHERE,
   * GROUP BY expressions and aggregate arguments they answer EVQL_ENOTSUP.  Type slot =
   * the (first) argument's. */
  EVQL_FAM_TO_STRING = 20,  /* to_string#string/X;  sql_tostring: NULL tag -> "NULL" */
  EVQL_FAM_CONCAT = 21,     /* concat / add#string/string;string; */
  EVQL_FAM_LCASE = 22,
  EVQL_FAM_UCASE = 23,
  EVQL_FAM_SUBSTRING = 24,  /* substring#string/string;int64; (1-based, negative from the end) */
  EVQL_FAM_LTRIM = 25,      /* leading / trailing ' ' only */
  EVQL_FAM_RTRIM = 26,
  EVQL_FAM_STARTSWITH = 27, /* -> bool */
  EVQL_FAM_ENDSWITH = 28
} evql_fn_family;
#define EVQL_FAM_LAST EVQL_FAM_ENDSWITH

#define EVQL_FN(family, type_slot) ((int64_t)(family) * 16 + (int64_t)(type_slot))

/*
 * Aggregate functions (X_CALL_INSTANCE).  count / sum_uint64 / sum_int64 are
 * the reference's live aggregates (expressions/aggregate.cc:35-219); the rest
 * are supplied by this build under the same SFunction vtable contract
 * (SFunction.h:41-86) because BASELINE.json's north_star asks for them
 * (SURVEY.md header, 8a a15).
 */
typedef enum {
  EVQL_AGG_NONE = 0,
  EVQL_AGG_COUNT = 1,       /* count#uint64/nil;      counts NULLs too      */
  EVQL_AGG_SUM_UINT64 = 2,  /* sum#uint64/uint64;     wraps mod 2^64        */
  EVQL_AGG_SUM_INT64 = 3,   /* sum#int64/int64;                             */
  EVQL_AGG_SUM_FLOAT64 = 4, /* sum#float64/float64;   build-supplied        */
  EVQL_AGG_MIN_UINT64 = 5,  /* min/max/mean skip STAG_NULL inputs (float min/max
                             * also NaN); empty => NULL */
  EVQL_AGG_MAX_UINT64 = 6,
  EVQL_AGG_MIN_INT64 = 7,
  EVQL_AGG_MAX_INT64 = 8,
  EVQL_AGG_MIN_FLOAT64 = 9,
  EVQL_AGG_MAX_FLOAT64 = 10,
  EVQL_AGG_MEAN_UINT64 = 11, /* -> float64 */
  EVQL_AGG_MEAN_INT64 = 12,
  EVQL_AGG_MEAN_FLOAT64 = 13,
  /* count_distinct#uint64/uint64; (aggregate.cc:77-137, std::set per group):
   * exact (HBM set of (group, value) pairs).  The pairs follow their groups through
   * evql_query_exchange and chain merges; evql_query_export_pairs / _import_pairs move them
   * for callers of export / import_groups; EVQL_MODE_PARTIAL rows carry the sorted values
   * (aggregate.cc:111-117), which evql_merge_* merges. */
  EVQL_AGG_COUNT_DISTINCT_UINT64 = 14
} evql_aggregate_fn;

/* X_CALL_INSTANCE arg0 */
#define EVQL_INSTANCE_ACCUMULATE 1
#define EVQL_INSTANCE_GET 2

typedef struct {
  uint32_t op;   /* evql_opcode */
  uint32_t argt; /* evql_stype of a literal / input */
  int64_t arg0;  /* CALL_PURE: EVQL_FN(..); CALL_INSTANCE: ACCUMULATE|GET;
                    LITERAL: byte offset into static_storage; INPUT: column
                    index; JUMP/CJUMP: target pc */
} evql_instr_t;

typedef struct {
  const evql_instr_t* code;
  uint32_t code_len;
  uint32_t method_call;       /* entry pc (vm::Program::method_call.offset) */
  uint32_t method_accumulate; /* entry pc; > 0 <=> aggregate program        */
  uint32_t return_type;       /* evql_stype */
  uint32_t aggregate_fn;      /* evql_aggregate_fn when method_accumulate > 0 */
  /* literal pool: each literal is its value bytes followed by one tag byte
   * (strings: u32 len, bytes, tag), i.e. the VM stack element layout */
  const uint8_t* static_storage;
  size_t static_storage_len;
} evql_program_t;

/* ------------------------------------------------------------------------ */
/* context                                                                    */
/* ------------------------------------------------------------------------ */
typedef struct evql_ctx evql_ctx_t;
typedef struct evql_table evql_table_t;
typedef struct evql_query evql_query_t;
typedef struct evql_writer evql_writer_t;

/* Binds to HIP device `device_ordinal`.  Fails with EVQL_EDEVICE when no GPU
 * is visible. `stream` may be NULL (a private stream is created) or an existing
 * hipStream_t (e.g. torch's current stream). */
int evql_ctx_create(int device_ordinal, void* stream, evql_ctx_t** out);
void evql_ctx_destroy(evql_ctx_t* ctx);
int evql_ctx_synchronize(evql_ctx_t* ctx);
void* evql_ctx_stream(evql_ctx_t* ctx);

/* ------------------------------------------------------------------------ */
/* tables: a cstable v0.2.0 file resident in HBM                              */
/* ------------------------------------------------------------------------ */

/* cstable::CSTableReader::openFile (io/cstable/cstable_reader.cc:133-200):
 * parse header / metablock / index, copy the page area to HBM. */
int evql_table_open_file(evql_ctx_t* ctx, const char* path, evql_table_t** out);

/* same for an in-memory image (reference: CSTableReader::openFile(arena),
 * cstable_reader.cc:202-230) */
int evql_table_open_image(evql_ctx_t* ctx, const void* image, size_t len,
                          evql_table_t** out);

void evql_table_close(evql_table_t* t);
uint64_t evql_table_num_rows(const evql_table_t* t);
int evql_table_num_columns(const evql_table_t* t);

typedef struct {
  char name[256];
  int32_t logical_type; /* evql_column_type */
  int32_t storage_type; /* evql_column_encoding */
  uint64_t column_id;
  uint32_t rlevel_max;
  uint32_t dlevel_max;
  uint32_t n_data_pages;
  uint32_t n_rlevel_pages;
  uint32_t n_dlevel_pages;
  uint64_t payload_bytes; /* algorithmic bytes of all streams */
} evql_column_info_t;

int evql_table_column_info(const evql_table_t* t, int idx,
                           evql_column_info_t* out);
/* size / copy-out of the image as it sits in HBM (tests of the generator) */
uint64_t evql_table_image_size(const evql_table_t* t);
int evql_table_download_image(const evql_table_t* t, void* dst, uint64_t len);

/*
 * Synthetic table generated directly into HBM in cstable v0.2.0 page layout
 * (the same bytes TableWriter would produce), SURVEY.md 8c(ii)/8d:
 *   x_i = xorshift64 (seed, shifts 13/7/17), stepped once per row
 *   k = x % k_mod, a = (x>>8)&0xffff, b = (x>>24)&0xffff,
 *   v = (x>>40)/1024.0, u = x % u_mod (0 = absent)
 * Columns present: bit0 k, bit1 a, bit2 b, bit3 v, bit4 u.  Encodings are
 * UINT64_PLAIN / FLOAT_IEEE754, or UINT32_BITPACKED of width k_bits for k
 * when k_bits > 0.
 */
typedef struct {
  uint64_t num_rows;
  uint64_t seed;
  uint64_t k_mod;
  uint64_t u_mod;
  uint32_t columns;
  uint32_t k_bits;
} evql_synth_spec_t;

int evql_table_generate(evql_ctx_t* ctx, const evql_synth_spec_t* spec,
                        evql_table_t** out);

/* ------------------------------------------------------------------------ */
/* host-side cstable writer (inputs for tests / loaders)                      */
/*   cstable::CSTableWriter (io/cstable/cstable_writer.cc:46-310)             */
/* ------------------------------------------------------------------------ */
typedef struct {
  const char* name;
  int32_t logical_type;
  int32_t storage_type;
  uint64_t column_id;
  uint32_t rlevel_max;
  uint32_t dlevel_max;
  uint32_t bitpack_max_value; /* 0 => reference default (0xffffffff / 1) */
} evql_column_spec_t;

int evql_writer_create(const evql_column_spec_t* cols, int ncols,
                       evql_writer_t** out);
/* bulk appends to one column.  rlvl/dlvl may be NULL (0 / dlevel_max);
 * present may be NULL (all present).  A slot with dlvl != dlevel_max (or
 * present == 0) is written as a NULL (ColumnWriter::writeNull). */
int evql_writer_put_uint(evql_writer_t* w, int col, uint64_t n,
                         const uint64_t* rlvl, const uint64_t* dlvl,
                         const uint8_t* present, const uint64_t* values);
int evql_writer_put_float(evql_writer_t* w, int col, uint64_t n,
                          const uint64_t* rlvl, const uint64_t* dlvl,
                          const uint8_t* present, const double* values);
int evql_writer_put_string(evql_writer_t* w, int col, uint64_t n,
                           const uint64_t* rlvl, const uint64_t* dlvl,
                           const uint8_t* present, const uint64_t* offsets,
                           const char* bytes);
int evql_writer_commit(evql_writer_t* w, uint64_t num_rows);
const void* evql_writer_image(const evql_writer_t* w, uint64_t* len);
/*
 * Re-encodes a cstable v0.1.0 image (io/cstable/columns/v1/, cstable.cc:89-132)
 * as v0.2.0; evql_table_open_* do this implicitly.  Host-only.  *out_len receives
 * the size needed; the image is written when dst_cap is large enough, otherwise
 * EVQL_EARG is returned with *out_len set (call with dst == NULL to size).
 */
int evql_cstable_upgrade(const void* image, uint64_t len, void* dst,
                         uint64_t dst_cap, uint64_t* out_len);
/*
 * The checks evql_table_open_* run before anything of the file reaches a kernel
 * (cstable::CSTableReader::openFile, io/cstable/cstable_reader.cc:133-200, plus the
 * page geometry the device code relies on): magic / version, SHA1-sealed metablock,
 * header, index within bounds (overflow-safe), every page of the size its encoding
 * fixes, required flat columns and definition-level streams large enough for
 * num_rows.  Host-only.  EVQL_EIO + evql_last_error() for a truncated or corrupt
 * file; the reference raises "end of column reached" when a scan gets that far.
 */
int evql_cstable_inspect(const void* image, uint64_t len, uint64_t* num_rows,
                         int* num_columns);
int evql_writer_write_file(const evql_writer_t* w, const char* path);
void evql_writer_destroy(evql_writer_t* w);

/* ------------------------------------------------------------------------ */
/* device-side cstable writer                                                  */
/*   the write side of compaction / result materialisation:                    */
/*   cstable::CSTableWriter::commitV2 (io/cstable/cstable_writer.cc:267-294),  */
/*   UInt64PageWriter / UInt32PageWriter / BitPackedIntPageWriter /            */
/*   LEB128PageWriter (io/cstable/columns/page_writer_*.cc), file index +      */
/*   metablock (io/cstable/cstable_file.cc:136-184)                            */
/* ------------------------------------------------------------------------ */
typedef struct {
  const uint64_t* values; /* DEVICE pointer: num_rows value words (u64 / f64 bits
                           * / bool as 0|1); entries of NULL rows are ignored */
  const uint8_t* nulls;   /* DEVICE pointer: num_rows bytes, 1 = NULL; given
                           * exactly for optional columns (dlevel_max == 1) */
  const uint8_t* bytes;   /* DEVICE pointer, STRING_PLAIN columns only: the byte heap;
                           * values[i] = (length << 40) | offset of string i in it
                           * (length < 2^24, offset < 2^40).  NULL otherwise */
  /* repeated / nested columns (rlevel_max > 0 or dlevel_max > 1), shredded as
   * RecordShredder does (io/cstable/RecordShredder.cc:113-176): one (r, d, value)
   * triple per SLOT, values[] / the level arrays num_slots long; a slot with
   * d != dlevel_max carries no value.  `nulls` stays NULL for such columns. */
  const uint8_t* rlevels; /* DEVICE pointer: num_slots bytes, or NULL when rlevel_max == 0 */
  const uint8_t* dlevels; /* DEVICE pointer: num_slots bytes */
  uint64_t num_slots;     /* 0 => one slot per row (flat columns) */
} evql_device_column_t;
/*
 * Encodes `ncols` SoA columns that sit in HBM into a cstable v0.2.0 image, in
 * HBM, and returns it as a table (evql_table_download_image / _write_file give
 * the file).  Required or optional (dlevel_max 1) columns in
 * UINT64_PLAIN, FLOAT_IEEE754, UINT32_PLAIN, UINT32_BITPACKED,
 * BOOLEAN_BITPACKED, UINT64_LEB128 or STRING_PLAIN (LenencStringPageWriter,
 * io/cstable/columns/page_writer_lenencstring.cc:37-69).  Repeated / nested columns
 * are given as level arrays + values per slot (ColumnWriter::write*(r, d, v),
 * io/cstable/ColumnWriter.cc:59-89).  Pages are placed in the order
 * PageManager::allocPage (io/cstable/page_manager.cc:50-74) would have handed them out
 * to a sequential writer, so the file is byte-identical to the one that writer (or
 * evql_writer_*) produces from the same values:
 *   EVQL_PAGE_ORDER_COLUMNS  the writer that appends one whole column after the other
 *   EVQL_PAGE_ORDER_ROWS     the writer that appends row by row / record by record,
 *                            every column per row in column order (RecordShredder,
 *                            io/cstable/RecordShredder.cc:113-176; ties inside one
 *                            record of a nested schema follow the column order)
 * evql_table_from_device_columns = EVQL_PAGE_ORDER_COLUMNS.
 */
typedef enum { EVQL_PAGE_ORDER_COLUMNS = 0, EVQL_PAGE_ORDER_ROWS = 1 } evql_page_order_t;
int evql_table_from_device_columns(evql_ctx_t* ctx, const evql_column_spec_t* cols,
                                   int ncols, const evql_device_column_t* data,
                                   uint64_t num_rows, evql_table_t** out);
int evql_table_from_device_columns_ordered(evql_ctx_t* ctx, const evql_column_spec_t* cols,
                                           int ncols, const evql_device_column_t* data,
                                           uint64_t num_rows, int page_order,
                                           evql_table_t** out);

/* ------------------------------------------------------------------------ */
/* the operator: GroupByExpression over FastCSTableScan / CSTableScan         */
/* ------------------------------------------------------------------------ */
typedef enum {
  /* GroupByExpression (groupby.cc:69-220): final values */
  EVQL_MODE_FINAL = 0,
  /* PartialGroupByExpression (groupby.cc:231-491): key + saved states */
  EVQL_MODE_PARTIAL = 1
} evql_group_mode;

typedef enum {
  /* FastCSTableScan (CSTableScan.cc:688-1009): one row per record, flat */
  EVQL_SCAN_FLAT = 0,
  /* CSTableScan, AggregationStrategy::NO_AGGREGATION (CSTableScan.cc:187-541):
   * Dremel assembly, one row per leaf repetition */
  EVQL_SCAN_NESTED = 1,
  /* CSTableScan, AggregationStrategy::AGGREGATE_WITHIN_RECORD_FLAT
   * (`sum(count(x) WITHIN RECORD)`, CSTableScan.cc:440-487): the scan select list
   * holds AGGREGATE programs; each accumulates over the rows of one record on which
   * the repetition level permits it (select_list_[i].rep_level >= cur_select_level_,
   * i.e. once per slot of its column) and the scan emits one row per record.
   * Lowered for count / sum over a bare column or a literal, without WHERE. */
  EVQL_SCAN_NESTED_WITHIN_RECORD = 2
} evql_scan_mode;

typedef struct {
  /* SequentialScanNode::selectedColumns(): X_INPUT(i) of `where` and
   * `scan_select` programs indexes this list */
  const char* const* scan_columns;
  const uint32_t* scan_column_types; /* evql_stype per scan column */
  uint32_t n_scan_columns;

  const evql_program_t* where; /* NULL => no predicate */

  /* scan select list; X_INPUT(j) of group_exprs / select_exprs indexes it */
  const evql_program_t* scan_select;
  uint32_t n_scan_select;

  const evql_program_t* group_exprs; /* n_group == 0 => one global group */
  uint32_t n_group;

  const evql_program_t* select_exprs;
  uint32_t n_select;

  /* AbstractCSTableScan::setFilter (CSTableScan.h:36-41): bit i == 0 drops
   * record i.  NULL => no external filter */
  const uint8_t* row_filter_bits;
  uint64_t row_filter_len; /* in rows */

  uint32_t group_mode; /* evql_group_mode */
  uint32_t scan_mode;  /* evql_scan_mode  */
  uint64_t groups_hint; /* expected number of groups; 0 = unknown (the reference's
                         * planner has no estimate): the first execute then aggregates
                         * a 256 Ki-row prefix, estimates the cardinality from the
                         * groups it finds and picks the LDS or the partitioned path */

  /* row range (partition slice) [row_begin, row_end); row_end == 0 => all */
  uint64_t row_begin;
  uint64_t row_end;

  /* sum#float64 (build-supplied): how the doubles are added up.
   *   EVQL_FLOAT_SUM_FAST  atomic double adds in whatever order the rows arrive:
   *                        within 1e-6 of the reference's row-order sum, not
   *                        bit-stable from run to run
   *   EVQL_FLOAT_SUM_EXACT every value is rounded once to a multiple of a fixed power
   *                        of two q (chosen so that 2^61 q covers float_sum_bound) and
   *                        the multiples are added as integers: the result does not
   *                        depend on the order of the rows, the run, the number of
   *                        workgroups or -- with the same bound on every partition --
   *                        on how the table is split over GPUs.  Error <= rows * q / 2.
   * float_sum_bound: an upper bound of |argument| over all rows; 0 = derived from the
   * table (maximum |value| of the columns the argument reads, through the
   * expression); EVQL_ENOTSUP when no finite bound can be derived.  A row beyond the
   * bound (or NaN / infinity) fails the query with EVQL_ERUNTIME. */
  uint32_t float_sum_mode;
  double float_sum_bound;
} evql_plan_desc_t;

#define EVQL_FLOAT_SUM_FAST 0
#define EVQL_FLOAT_SUM_EXACT 1

/* Lowers the plan to a fused kernel and compiles it (cached by fingerprint).
 * Replaces DefaultScheduler::buildGroupByExpression + buildSequentialScan
 * (sql/scheduler.cc:134-182).  EVQL_ENOTSUP => not lowerable. */
int evql_query_create(evql_ctx_t* ctx, evql_table_t* table,
                      const evql_plan_desc_t* plan, evql_query_t** out);
void evql_query_destroy(evql_query_t* q);

/* txn_->triggerHeartbeat() (transaction.cc:54-60): return non-zero to abort */
typedef int (*evql_heartbeat_fn)(void* user);

/* TableExpression::execute (table_expression.h:38): runs scan+filter+aggregate
 * to completion on the device.  `hb` is called before the launch, every 5 ms while the
 * kernels run (the host polls the stream) and once after them -- the reference beats
 * once per input batch (groupby.cc:100-105); non-zero = stop: honoured when the
 * kernels have drained, execute then fails with EVQL_ERUNTIME "query aborted". */
int evql_query_execute(evql_query_t* q, evql_heartbeat_fn hb, void* user);

/* asynchronous form used by benchmarks: enqueue the kernels on the context
 * stream without waiting; evql_query_finish() waits and collects the result. */
int evql_query_launch(evql_query_t* q);
int evql_query_finish(evql_query_t* q);

/* TableExpression::getColumnCount / getColumnType (table_expression.h:44-46);
 * valid right after evql_query_create (ResultCursor sizes its buffers before
 * execute(), result_cursor.cc:39-42). */
int evql_query_column_count(const evql_query_t* q);
int evql_query_column_type(const evql_query_t* q, int idx);

typedef struct {
  const uint8_t* data; /* packed SVector elements (svalue.cc:410-517) */
  size_t size;         /* bytes */
} evql_column_buf_t;

/* TableExpression::nextBatch (table_expression.h:40-42): up to max_rows (the
 * reference uses kOutputBatchSize = 1024) result rows; cols[i] receive the
 * packed bytes to SVector::append.  Buffers stay valid until the next call.
 * *nrows == 0 => EOF (and stays 0). */
int evql_query_next_batch(evql_query_t* q, size_t max_rows,
                          evql_column_buf_t* cols, size_t* nrows);

/* statistics of the last execute */
typedef struct {
  uint64_t rows_scanned;
  uint64_t rows_passed;
  uint64_t num_groups;
  uint64_t algorithmic_bytes; /* SURVEY 8d B_alg of the columns referenced */
  double kernel_ms;           /* device time of the dominant (scan) kernel */
  double total_ms;            /* all kernels of the query                   */
  uint32_t n_kernel_launches;
  uint32_t used_lds_table;
  uint64_t estimated_groups; /* plans without groups_hint: what the sample pass of the
                              * first execute estimated (0 = no estimate made) */
} evql_query_stats_t;
int evql_query_stats(const evql_query_t* q, evql_query_stats_t* out);

/* the generated HIP source of the fused kernel (inspection / tests) */
const char* evql_query_kernel_source(const evql_query_t* q);

/* ------------------------------------------------------------------------ */
/* partial aggregates across partitions / GPUs                                */
/*   PartialGroupByExpression -> GroupByMergeExpression, groupby.cc:438-615   */
/* ------------------------------------------------------------------------ */

/* After execute() in EVQL_MODE_PARTIAL the group table stays in HBM as SoA
 * device arrays: words[w * capacity + slot], w < words_per_group.  Word 0 is
 * the slot state/identity, the following words the key and aggregate states.
 * These are what ranks exchange over RCCL. */
typedef struct {
  void* device_words;    /* uint64_t[words_per_group * capacity] in HBM */
  uint64_t capacity;     /* slots */
  uint32_t words_per_group;
  uint64_t num_groups;
} evql_partial_view_t;
int evql_query_partial_view(evql_query_t* q, evql_partial_view_t* out);

/* number of 8-byte words in one exported group record, and compaction of the
 * table into dense records (record-major) for the wire */
int evql_query_export_groups(evql_query_t* q, void* device_dst,
                             uint64_t max_groups, uint64_t* n_groups);
/* merge dense records produced by evql_query_export_groups on another
 * partition / rank into this query's table (mergeInstance per aggregate); the table
 * is regrown first when the incoming groups would not fit.  Both answer EVQL_ENOTSUP
 * for plans whose records name a first ROW (string / multi-column keys, non-aggregate
 * select expressions) -- a row index means nothing outside the table that produced
 * it; evql_query_exchange carries the values themselves.  count_distinct: see
 * evql_query_export_pairs. */
int evql_query_import_groups(evql_query_t* q, const void* device_src,
                             uint64_t n_groups);
uint32_t evql_query_record_words(const evql_query_t* q);
/* count_distinct (aggregate.cc:77-137: a std::set per group, merged by insertion) for
 * callers that move partial aggregates themselves: the stored (group, value) pairs of the
 * plan's `which`-th count_distinct aggregate as 3-word triples.  _import_pairs inserts them
 * into this query's set and adds 1 to the aggregate of every group a pair is new to; call
 * it AFTER evql_query_import_groups of the records the pairs belong to (import_groups
 * leaves count_distinct words alone: the counts follow from the pairs). */
uint32_t evql_query_distinct_aggregates(const evql_query_t* q);
int evql_query_export_pairs(evql_query_t* q, uint32_t which, void* device_dst,
                            uint64_t max_pairs, uint64_t* n_pairs);
int evql_query_import_pairs(evql_query_t* q, uint32_t which, const void* device_src,
                            uint64_t n_pairs);
/* empties the query's group table without scanning, so that it can serve as the
 * merge target of partial aggregates (GroupByMergeExpression, groupby.cc:528-637) */
int evql_query_reset(evql_query_t* q);

/* ------------------------------------------------------------------------ */
/* the exchange step between the GPUs of a node                               */
/*   the reference fans partial aggregates in over TCP: PartialGroupByExpression */
/*   rows -> EVQL_OP_QUERY_PARTIALAGGR frames -> GroupByMergeExpression          */
/*   (groupby.cc:438-472, 528-637; server/sql/scheduler.cc:117-162).  Here every  */
/*   partition's groups sit in the HBM of the GPU that scanned it and travel as   */
/*   device buffers.                                                              */
/* ------------------------------------------------------------------------ */
/*
 * A transport moves device memory between the ranks (one rank = one evql_ctx = one
 * GPU).  Built in: RCCL over xGMI for one process per GPU (evql_exchange_create_rccl)
 * and an in-process hub for one process driving several contexts from one thread per
 * rank (evql_hub_*; the EventQL server itself runs partitions as threads of one
 * process).  Anything else can be plugged in through the callbacks.  A callback
 * returns 0 or an evql_status.
 */
typedef struct {
  void* user;
  /* every rank contributes n words of HOST memory; recv gets nranks * n words in
   * rank order */
  int (*all_gather_u64)(void* user, const uint64_t* send, uint64_t n, uint64_t* recv);
  /* variable all-to-all of DEVICE memory in 8-byte words: send_counts[r] words go to
   * rank r from d_send (packed in rank order), recv_counts[r] words arrive from rank r
   * in d_recv (packed in rank order).  d_send is complete on `hip_stream`; d_recv
   * must be complete on it (or the call synchronous) on return */
  int (*all_to_all_words)(void* user, const uint64_t* d_send, const uint64_t* send_counts,
                          uint64_t* d_recv, const uint64_t* recv_counts, void* hip_stream);
  const char* name; /* "rccl", "hub", ... reported by evql_exchange_backend */
} evql_transport_t;

typedef struct evql_exchange evql_exchange_t;
typedef struct evql_hub evql_hub_t;

int evql_exchange_create(evql_ctx_t* ctx, int nranks, int rank, const evql_transport_t* transport,
                         evql_exchange_t** out);
/* RCCL: rank 0 makes the id (128 bytes, ncclUniqueId) and hands it to the others out
 * of band (the launcher's rendezvous); collective: every rank calls _create_rccl */
int evql_rccl_unique_id(void* id128);
int evql_exchange_create_rccl(evql_ctx_t* ctx, int nranks, int rank, const void* id128,
                              evql_exchange_t** out);
/* in-process hub: nranks contexts (same or different devices) in ONE process, every
 * rank's calls on its own thread; copies are device-to-device (peer) copies */
int evql_hub_create(int nranks, evql_hub_t** out);
void evql_hub_destroy(evql_hub_t* hub);
int evql_exchange_create_hub(evql_ctx_t* ctx, evql_hub_t* hub, int rank, evql_exchange_t** out);
void evql_exchange_destroy(evql_exchange_t* x);
const char* evql_exchange_backend(const evql_exchange_t* x);

typedef enum {
  /* low cardinality: every rank ends up with every group (the reference's
   * coordinator; any rank may emit the result) */
  EVQL_EXCHANGE_GATHER_ALL = 0,
  /* high cardinality: group records are bucketed on the device by owner =
   * hash(identity) % nranks; every rank ends up with the groups it owns and emits
   * them through next_batch -- the result stays distributed */
  EVQL_EXCHANGE_BY_OWNER = 1
} evql_exchange_mode;

/*
 * Collective, after execute() on every rank (same plan everywhere, each over its own
 * partition): exports this rank's groups, exchanges them and merges what arrives
 * (mergeInstance per aggregate, groupby.cc:577-612) -- batches in RANK ORDER into a
 * fresh table, so that a float sum is added up in the same order on every rank and
 * in every run of the same partial aggregates (record sets of 2^18 records and more are
 * merged bucket by bucket in the LDS instead -- no scattered HBM accesses; there only
 * EVQL_FLOAT_SUM_EXACT sums are independent of the order).  Plans that read first-row values
 * (string / multi-column keys, non-aggregate select expressions) carry them in the
 * records: the first row of the lowest rank that has the group wins, strings travel
 * as bytes.  next_batch then yields the merged groups.  count_distinct: the (group,
 * value) pairs follow their groups and are counted again in the merged set
 * (aggregate.cc:119-137), also for EVQL_MODE_PARTIAL plans (their rows then carry the
 * merged sets' values).
 */
int evql_query_exchange(evql_query_t* q, evql_exchange_t* x, int mode);

typedef struct {
  uint64_t groups_sent;     /* records this rank exported */
  uint64_t groups_received; /* records merged here (own ones included) */
  uint64_t bytes_sent;      /* record + string bytes that left this rank */
  double export_ms, transfer_ms, merge_ms;
  uint64_t merge_buckets;   /* != 0: merged in the LDS, bucket by bucket (large record sets);
                               0: merged through an HBM table, one launch per source rank */
} evql_exchange_stats_t;
int evql_exchange_last_stats(const evql_exchange_t* x, evql_exchange_stats_t* out);

/*
 * GroupByMergeExpression over partial aggregates that arrive as BYTES
 * (groupby.cc:493-672): the coordinator's merge of PartialGroupBy rows from
 * other nodes -- this build's EVQL_MODE_PARTIAL output or an unmodified
 * reference node's.  Only plan->select_exprs is read.  Host-only (sequential
 * LEB128 / SValue decoding, O(groups) work); partial tables resident on other
 * GPUs are merged on the device with evql_query_import_groups instead.
 *   add_frame: payload of one EVQL_OP_QUERY_PARTIALAGGR_RESULT frame
 *              (varuint flags, varuint count, count x {20-B key, states...},
 *              frames/query_partialaggr_result.cc:53-57)
 *   add_rows:  the two STRING SVectors (key, data) of a PartialGroupBy
 *              nextBatch, as evql_query_next_batch returns them
 *   next_batch: as evql_query_next_batch (method_call per select expression;
 *              non-aggregates carry the value decoded last, groupby.cc:606-610)
 */
typedef struct evql_merge evql_merge_t;
int evql_merge_create(const evql_plan_desc_t* plan, evql_merge_t** out);
void evql_merge_destroy(evql_merge_t* m);
int evql_merge_add_frame(evql_merge_t* m, const void* payload, size_t len);
int evql_merge_add_rows(evql_merge_t* m, const void* keys, size_t keys_len,
                        const void* data, size_t data_len, size_t nrows);
uint64_t evql_merge_num_groups(const evql_merge_t* m);
size_t evql_merge_column_count(const evql_merge_t* m);
uint32_t evql_merge_column_type(const evql_merge_t* m, size_t idx);
int evql_merge_next_batch(evql_merge_t* m, size_t max_rows,
                          evql_column_buf_t* cols, size_t* nrows);

/* ------------------------------------------------------------------------ */
/* ORDER BY .. LIMIT above the GROUP BY                                       */
/*   OrderByExpression  sql/statements/select/orderby.cc:60-160               */
/*   LimitExpression    sql/statements/select/limit.cc:52-125                 */
/* ------------------------------------------------------------------------ */
/*
 * Fuses LimitExpression(limit, offset, OrderByExpression(specs, <this query>))
 * into the operator so that a high-cardinality GROUP BY does not ship every
 * group through nextBatch: the device selects the offset+limit smallest records
 * by the first sort key (radix select over the group table), only those are
 * fetched, and the host orders them by all specs.  Sort expressions are programs
 * over the query's OUTPUT columns (X_INPUT i = select expression i), compared with
 * the reference's cmp#int64/X;X; comparators: payloads only, NULL reads as 0, ties
 * in unspecified order (the reference uses std::sort).  limit < 0: ORDER BY
 * only.  limit == 0 yields no rows (limit.cc:58).  Call before execute().
 * EVQL_ENOTSUP when the first sort expression is not a select column whose value
 * the device can read from a group record (the group key, or a bare count / sum /
 * min / max / mean): the caller then keeps the CPU operators above this one.
 */
typedef struct {
  evql_program_t expr;
  uint32_t descending;
} evql_sort_spec_t;
int evql_query_set_order(evql_query_t* q, const evql_sort_spec_t* specs,
                         uint32_t n_specs, int64_t limit, uint64_t offset);

/* ------------------------------------------------------------------------ */
/* LSM row filters (PartitionCursor::openNextTable,                           */
/* server/sql/partition_cursor.cc:83-226)                                     */
/* ------------------------------------------------------------------------ */
/*
 * A partition is a chain of cstables scanned newest first: head arena,
 * compacting arena, then the LSM files from the newest to the oldest.  A row is
 * scanned unless it is skipped (`__lsm_skip`, or the arena's skiplist) or a row
 * in front of it with the same `__lsm_id` was a kept update
 * (`__lsm_is_update`).  Tables must be added in scan order; `flags` carry the file's
 * LSMTableRef bits (db/partition_state.proto: has_skiplist, has_updates), which decide
 * -- exactly as partition_cursor.cc:149-155 does -- whether a file gets a filter at all:
 *     !has_skiplist && <oldest file> && <no update remembered yet>    => none
 *     !has_skiplist && !has_updates  && <no update remembered yet>    => none
 * (a file without a filter is scanned whole and its updates are not remembered).
 * arena_skiplist: one byte per row (PartitionArena::SkiplistReader::readNext) for the
 * two arenas, NULL for files.  _build computes every table's filter on the device;
 * _filter hands out a host copy of the bitmap (bit r of byte r/8 set <=> row r is
 * scanned) in the layout evql_plan_desc_t::row_filter_bits takes, valid until _destroy
 * or the next _build; *bits == NULL: the table needs no filter (setFilter is not
 * called, partition_cursor.cc:215-217).  evql_query_create_chain reads the filters
 * where they are, in HBM.
 * An id that is not 20 bytes long fails with EVQL_ERUNTIME "invalid SHA1Hash"
 * (util/SHA1.cc:79-85).
 */
#define EVQL_LSM_HAS_SKIPLIST 1u /* LSMTableRef::has_skiplist: read __lsm_skip */
#define EVQL_LSM_HAS_UPDATES 2u  /* LSMTableRef::has_updates */
typedef struct evql_lsm_chain evql_lsm_chain_t;
int evql_lsm_chain_create(evql_ctx_t* ctx, evql_lsm_chain_t** out);
void evql_lsm_chain_destroy(evql_lsm_chain_t* ch);
int evql_lsm_chain_add(evql_lsm_chain_t* ch, evql_table_t* table, uint32_t flags,
                       const uint8_t* arena_skiplist, uint64_t arena_skiplist_len);
int evql_lsm_chain_build(evql_lsm_chain_t* ch);
int evql_lsm_chain_length(const evql_lsm_chain_t* ch);
int evql_lsm_chain_filter(evql_lsm_chain_t* ch, int idx,
                          const uint8_t** bits, uint64_t* nrows,
                          uint64_t* rows_kept);

/*
 * The operator over a whole partition: GroupByExpression (or PartialGroupByExpression)
 * whose input is PartitionCursor (server/sql/partition_cursor.cc:34-235) -- the scans of
 * the chain's tables one after the other, newest first, each under its row filter
 * (AbstractCSTableScan::setFilter, sql/CSTableScan.h:36-41), feeding ONE group map.
 * Here every table is scanned by its own fused kernel launch (tables of one partition
 * may differ in encodings) and the per-table groups are merged on the device in chain
 * order, so that non-aggregate select expressions keep the value of the group's first
 * row in scan order (groupby.cc:161-172) and count_distinct counts the union of the
 * tables' sets.  `ch` must be built and outlive the query; every other entry point
 * (execute, next_batch, set_order, stats ...) takes the returned query as usual.
 * EVQL_SCAN_FLAT only (PartitionCursor builds FastCSTableScan for NO_AGGREGATION
 * statements, :197-204; anything else answers EVQL_ENOTSUP).
 */
int evql_query_create_chain(evql_ctx_t* ctx, evql_lsm_chain_t* ch,
                            const evql_plan_desc_t* plan, evql_query_t** out);

/* ------------------------------------------------------------------------ */
/* build support                                                              */
/* ------------------------------------------------------------------------ */
/* Compile the fused kernel of `plan` for gfx950 without a device (used by
 * __graft_entry__.build() and the CPU test-suite).  Stores the code object in
 * the on-disk kernel cache when cache_dir != NULL. */
int evql_compile_only(const evql_plan_desc_t* plan,
                      const evql_column_info_t* columns, int ncolumns,
                      const char* cache_dir, size_t* code_size);
/* on-disk cache of the compiled plan kernels (code objects named by the fingerprint of
 * their source).  Default: the directory `_kcache` next to this shared library, shared by
 * every process that loads it; "" switches the disk cache off. */
void evql_set_kernel_cache_dir(const char* dir);

#ifdef __cplusplus
}
#endif
#endif /* EVQL_GPU_H */
