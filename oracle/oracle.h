/*
 * oracle.h -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * A plain-C CPU restatement of the reference algorithm for the hot path
 * (cstable decode -> FastCSTableScan/CSTableScan -> VM -> GroupByExpression),
 * written from the reference sources, each function citing the file:line it
 * follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load liboracle.so; the product (eventql_amd/, include/) never does.
 *
 * Parity pins (see DESIGN.md "Oracle"):
 *   - decode half: checked against the reference's own cstable library
 *     compiled in place (oracle/_ref, `make ref`) and against
 *     test/sql_testdata/testtbl.cst
 *   - csql half: checked against the reference's own csql engine compiled in place
 *     (oracle/ref_csql/build.sh -> oracle/_ref/csql_probe): results, PartialGroupBy
 *     bytes and compiled bytecode of 587 queries incl. GROUP BYs over
 *     eventql::PartitionCursor, committed as tests/golden/ref_csql_*.json; plus the
 *     reference's test fixtures (test/sql/00001,00002,00014; Runtime_test.cc)
 *   - unpinned (absent from this snapshot of the reference): sum(float64), min, max,
 *     mean; WITHIN RECORD beyond the three Runtime_test.cc answers
 */
#ifndef EVQL_ORACLE_H
#define EVQL_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/evql_gpu.h" /* struct layouts of the plan descriptor only */

#ifdef __cplusplus
extern "C" {
#endif

const char* orc_last_error(void);

/* ---- sha1.c (reference: util/SHA1.cc) ---------------------------------- */
void orc_sha1(const void* data, size_t len, uint8_t out[20]);

/* ---- cstable_oracle.c --------------------------------------------------- */
typedef struct orc_table orc_table_t;
typedef struct orc_column orc_column_t;

orc_table_t* orc_table_open(const char* path);
orc_table_t* orc_table_open_image(const void* image, size_t len);
void orc_table_close(orc_table_t* t);
int orc_table_version(const orc_table_t* t); /* 1 or 2 */
uint64_t orc_table_num_rows(const orc_table_t* t);
int orc_table_num_columns(const orc_table_t* t);
int orc_table_column_info(const orc_table_t* t, int idx, char* name_out,
                          int* logical_type, int* storage_type,
                          uint64_t* column_id, uint64_t* rlevel_max,
                          uint64_t* dlevel_max);
/* v0.1.0 only: total number of (r,d,value) slots of a column; 0 for v0.2.0 */
uint64_t orc_table_column_num_values(const orc_table_t* t, const char* name);

/* sequential column cursor = cstable::ColumnReader */
orc_column_t* orc_column_open(orc_table_t* t, const char* name);
void orc_column_close(orc_column_t* c);
uint64_t orc_column_next_rlevel(orc_column_t* c);
int orc_column_read_uint(orc_column_t* c, uint64_t n, uint64_t* rl,
                         uint64_t* dl, uint8_t* present, uint64_t* v);
int orc_column_read_float(orc_column_t* c, uint64_t n, uint64_t* rl,
                          uint64_t* dl, uint8_t* present, double* v);
int orc_column_read_string(orc_column_t* c, uint64_t n, uint64_t* rl,
                           uint64_t* dl, uint8_t* present, uint64_t* offsets,
                           char* bytes, uint64_t cap);

int orc_column_read_string_alloc(orc_column_t* c, uint64_t* r, uint64_t* d,
                                 uint8_t* present, char** buf, uint64_t* cap,
                                 uint64_t* len);
uint32_t orc_column_rmax(const orc_column_t* c);
uint32_t orc_column_dmax(const orc_column_t* c);
int orc_column_logical_type(const orc_column_t* c);

/* ---- csql_oracle.c ------------------------------------------------------- */
typedef struct orc_result orc_result_t;

/* Runs the whole operator tree GroupByExpression(FastCSTableScan | CSTableScan)
 * (or the bare scan when plan->n_select == 0 && plan->n_group == 0) on one
 * thread, exactly in reference order.  Returns NULL on error. */
orc_result_t* orc_query_run(orc_table_t* t, const evql_plan_desc_t* plan);
/* the same operator tree over PartitionCursor (server/sql/partition_cursor.cc:56-81):
 * the flat scans of `ntables` tables one after the other, table i under the row filter
 * filters[i] (bit r of byte r/8; NULL = none), one group map across all of them */
orc_result_t* orc_query_run_chain(orc_table_t* const* tables, int ntables,
                                  const uint8_t* const* filters,
                                  const uint64_t* filter_lens,
                                  const evql_plan_desc_t* plan);
void orc_result_free(orc_result_t* r);
int orc_result_num_columns(const orc_result_t* r);
int orc_result_column_type(const orc_result_t* r, int col);
uint64_t orc_result_num_rows(const orc_result_t* r);
/* all rows of one output column as packed SVector bytes (svalue.cc:410-517) */
const uint8_t* orc_result_column_data(const orc_result_t* r, int col,
                                      size_t* size);
/* EVQL_MODE_PARTIAL: 20-byte SHA1 group keys (row-major) */
const uint8_t* orc_result_group_keys(const orc_result_t* r);
uint64_t orc_result_rows_scanned(const orc_result_t* r);
uint64_t orc_result_rows_passed(const orc_result_t* r);
const char* orc_query_error(void);

/* OrderByExpression (orderby.cc:60-160) then LimitExpression (limit.cc:52-125)
 * applied in place to a result; sort programs index the result's columns.
 * limit < 0: no LIMIT.  Ties keep their input order.  Returns 0 / -1. */
int orc_result_order_limit(orc_result_t* r, const evql_sort_spec_t* specs,
                           uint32_t n_specs, int64_t limit, uint64_t offset);

/* ---- lsm_oracle.c ---------------------------------------------------------- */
/* PartitionCursor::openNextTable row filters (partition_cursor.cc:160-195):
 * call once per table of the chain, newest first; filter_out receives one byte
 * (0 dropped / 1 scanned) per row.  Returns 0, -1 on a read error, -2 for an id
 * that is not 20 bytes ("invalid SHA1Hash"). */
typedef struct orc_lsm orc_lsm_t;
orc_lsm_t* orc_lsm_create(void);
void orc_lsm_free(orc_lsm_t* m);
int orc_lsm_next_table(orc_lsm_t* m, orc_table_t* t, int has_skip_column,
                       const uint8_t* arena_skip, uint8_t* filter_out);
/* the same step with the file's LSMTableRef flags and the two "no filter needed"
 * shortcuts of partition_cursor.cc:149-155 (see lsm_oracle.c) */
int orc_lsm_next_file(orc_lsm_t* m, orc_table_t* t, int has_skiplist, int has_updates,
                      int is_oldest, const uint8_t* arena_skip, uint8_t* filter_out,
                      int* needs_filter_out);

/* GroupByMergeExpression (groupby.cc:528-672) over `nframes` partial-aggregate
 * frame payloads (varuint flags, varuint count, rows); only plan->select_exprs
 * is read.  orc_result_group_keys() returns the merged groups' SHA1 keys. */
orc_result_t* orc_merge_frames(const evql_plan_desc_t* plan,
                               const uint8_t* const* frames,
                               const size_t* lens, int nframes);

#ifdef __cplusplus
}
#endif
#endif
