/* lsm_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * The row-filter loop of PartitionCursor::openNextTable restated,
 * server/sql/partition_cursor.cc:160-195: tables are visited newest first and an
 * id set carries over from table to table;
 *
 *     id = SHA1Hash(__lsm_id)            -- raises unless 20 bytes (util/SHA1.cc:79-85)
 *     is_update = __lsm_is_update; skip = __lsm_skip (if the file has a skiplist)
 *     if the table is an arena: skip = arena skiplist
 *     if (skip || id_set.count(id)) filter[i] = false;
 *     else { if (is_update) id_set.insert(id); filter[i] = true; }
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

struct orc_lsm {
  uint8_t* ids; /* 20 bytes each */
  size_t n, cap;
  int64_t* slots;
  size_t nslots;
};

static uint64_t id_hash(const uint8_t* k) {
  uint64_t h;
  memcpy(&h, k, 8);
  return h * 0x9e3779b97f4a7c15ull;
}

static void set_grow(orc_lsm_t* m) {
  size_t ns = m->nslots ? m->nslots * 2 : 1024;
  int64_t* s = (int64_t*) malloc(ns * sizeof(int64_t));
  for (size_t i = 0; i < ns; ++i) s[i] = -1;
  for (size_t i = 0; i < m->n; ++i) {
    size_t p = (id_hash(m->ids + 20 * i) >> 20) & (ns - 1);
    while (s[p] >= 0) p = (p + 1) & (ns - 1);
    s[p] = (int64_t) i;
  }
  free(m->slots);
  m->slots = s;
  m->nslots = ns;
}

static int set_contains(const orc_lsm_t* m, const uint8_t* id) {
  if (!m->nslots) return 0;
  size_t p = (id_hash(id) >> 20) & (m->nslots - 1);
  while (m->slots[p] >= 0) {
    if (memcmp(m->ids + 20 * m->slots[p], id, 20) == 0) return 1;
    p = (p + 1) & (m->nslots - 1);
  }
  return 0;
}

static void set_insert(orc_lsm_t* m, const uint8_t* id) {
  if ((m->n + 1) * 2 > m->nslots) set_grow(m);
  if (m->n == m->cap) {
    m->cap = m->cap ? m->cap * 2 : 1024;
    m->ids = (uint8_t*) realloc(m->ids, m->cap * 20);
  }
  memcpy(m->ids + 20 * m->n, id, 20);
  size_t p = (id_hash(id) >> 20) & (m->nslots - 1);
  while (m->slots[p] >= 0) p = (p + 1) & (m->nslots - 1);
  m->slots[p] = (int64_t) m->n;
  m->n++;
}

orc_lsm_t* orc_lsm_create(void) { return (orc_lsm_t*) calloc(1, sizeof(orc_lsm_t)); }

void orc_lsm_free(orc_lsm_t* m) {
  if (!m) return;
  free(m->ids);
  free(m->slots);
  free(m);
}

int orc_lsm_next_table(orc_lsm_t* m, orc_table_t* t, int has_skip_column,
                       const uint8_t* arena_skip, uint8_t* filter_out) {
  /* (a table that is always filtered: an arena, or a file that may hold updates and
   * is not the oldest one) */
  return orc_lsm_next_file(m, t, has_skip_column, 1, 0, arena_skip, filter_out, NULL);
}

/* One step of PartitionCursor::openNextTable for an LSM file (or an arena when
 * arena_skip != NULL), partition_cursor.cc:134-195.  `has_skiplist` / `has_updates`
 * are the LSMTableRef flags of the file (db/partition_state.proto), `is_oldest` says
 * tblidx == 0.  A file needs no filter -- every row is scanned and, NOTE, its updates
 * are not remembered either -- when (:149-155)
 *     !has_skiplist && tblidx == 0 && id_set.empty()
 *  || !has_skiplist && !has_updates && id_set.empty()
 * *needs_filter_out receives whether setFilter was called (:215-217). */
int orc_lsm_next_file(orc_lsm_t* m, orc_table_t* t, int has_skiplist, int has_updates,
                      int is_oldest, const uint8_t* arena_skip, uint8_t* filter_out,
                      int* needs_filter_out) {
  const uint64_t n = orc_table_num_rows(t);
  int needs_filter = 1;
  if (!arena_skip) {
    if (!has_skiplist && is_oldest && m->n == 0) needs_filter = 0;
    if (!has_skiplist && !has_updates && m->n == 0) needs_filter = 0;
  }
  if (needs_filter_out) *needs_filter_out = needs_filter;
  if (!needs_filter) {
    memset(filter_out, 1, n); /* std::vector<bool> filter(numRecords, true), :161 */
    return 0;
  }
  const int has_skip_column = has_skiplist && !arena_skip;
  orc_column_t* id_col = orc_column_open(t, "__lsm_id");
  orc_column_t* upd_col = orc_column_open(t, "__lsm_is_update");
  orc_column_t* skip_col = has_skip_column ? orc_column_open(t, "__lsm_skip") : NULL;
  int rc = 0;
  if (!id_col || !upd_col || (has_skip_column && !skip_col)) rc = -1;
  char* buf = NULL;
  uint64_t cap = 0;
  for (uint64_t i = 0; i < n && rc == 0; ++i) {
    uint64_t r, d, len = 0, v = 0;
    uint8_t present;
    if (orc_column_read_string_alloc(id_col, &r, &d, &present, &buf, &cap, &len)) {
      rc = -1;
      break;
    }
    if (len != 20) { /* "invalid SHA1Hash" */
      rc = -2;
      break;
    }
    if (orc_column_read_uint(upd_col, 1, &r, &d, &present, &v)) {
      rc = -1;
      break;
    }
    const int is_update = v > 0; /* readBoolean, column_reader_uint.cc:78-90 */
    int skip = 0;
    if (skip_col) {
      if (orc_column_read_uint(skip_col, 1, &r, &d, &present, &v)) {
        rc = -1;
        break;
      }
      skip = v > 0;
    }
    if (arena_skip) skip = arena_skip[i] != 0;
    if (skip || set_contains(m, (const uint8_t*) buf)) {
      filter_out[i] = 0;
    } else {
      if (is_update) set_insert(m, (const uint8_t*) buf);
      filter_out[i] = 1;
    }
  }
  free(buf);
  if (id_col) orc_column_close(id_col);
  if (upd_col) orc_column_close(upd_col);
  if (skip_col) orc_column_close(skip_col);
  return rc;
}
