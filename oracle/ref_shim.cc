/*
 * ref_shim.cc -- TEST INFRASTRUCTURE ONLY.
 *
 * A thin extern "C" veneer over the *reference's own* cstable library
 * (compiled from the sources where they lie under /root/reference by
 * oracle/Makefile, target `ref`).  It exposes the reference CSTableWriter and
 * CSTableReader so that tests can
 *   (a) read files produced by eventql_amd's writer with the reference reader,
 *   (b) produce files with the reference writer and decode them with the
 *       oracle restatement and the HIP path,
 *   (c) time the reference's decode loop as a CPU baseline.
 * Nothing here is shipped; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load oracle/_ref/libcstable_ref.so.
 *
 * Reference interfaces wrapped:
 *   cstable::CSTableWriter::createFile / getColumnWriter / addRows / commit
 *       src/eventql/io/cstable/cstable_writer.cc:46-82, 267-310
 *   cstable::CSTableReader::openFile / getColumnReader / numRecords / columns
 *       src/eventql/io/cstable/cstable_reader.cc:133-315
 *   cstable::ColumnReader::readUnsignedInt / readFloat / readString
 *       src/eventql/io/cstable/ColumnReader.h:35-83
 *   cstable::TableSchema::addColumn / addSubrecord(Array)
 *       src/eventql/io/cstable/TableSchema.h
 */
#include <eventql/io/cstable/cstable_writer.h>
#include <eventql/io/cstable/cstable_reader.h>
#include <eventql/io/cstable/TableSchema.h>
#include <eventql/util/SHA1.h>
#include <string.h>
#include <string>
#include <vector>
#include <memory>

using namespace cstable;

namespace {
thread_local std::string g_err;

struct SchemaNode {
  std::string name;
  int type;       // cstable::ColumnType
  int encoding;   // cstable::ColumnEncoding
  int repeated;
  int optional;
  int parent;     // -1 = root
};

static void buildSchema(
    const std::vector<SchemaNode>& nodes,
    int parent,
    TableSchema* out) {
  for (size_t i = 0; i < nodes.size(); ++i) {
    const auto& n = nodes[i];
    if (n.parent != parent) continue;
    if (n.type == (int) ColumnType::SUBRECORD) {
      TableSchema sub;
      buildSchema(nodes, (int) i, &sub);
      if (n.repeated) {
        out->addSubrecordArray(n.name, sub, n.optional);
      } else {
        out->addSubrecord(n.name, sub, n.optional);
      }
    } else {
      out->addColumn(
          n.name,
          (ColumnType) n.type,
          (ColumnEncoding) n.encoding,
          n.repeated,
          n.optional);
    }
  }
}

struct RefWriter {
  RefPtr<CSTableWriter> w;
};

struct RefReader {
  RefPtr<CSTableReader> r;
};
}  // namespace

#define REF_TRY try {
#define REF_CATCH(rv)                           \
  } catch (const std::exception& e) {           \
    g_err = e.what();                           \
    return rv;                                  \
  } catch (...) {                               \
    g_err = "unknown exception";                \
    return rv;                                  \
  }

extern "C" {

const char* ref_last_error() { return g_err.c_str(); }

/* schema given as parallel arrays; parent[i] = index of enclosing SUBRECORD or -1 */
void* ref_writer_create(
    const char* path,
    int nnodes,
    const char* const* names,
    const int* types,
    const int* encodings,
    const int* repeated,
    const int* optional,
    const int* parent) {
  REF_TRY
  std::vector<SchemaNode> nodes;
  for (int i = 0; i < nnodes; ++i) {
    nodes.push_back(
        {names[i], types[i], encodings[i], repeated[i], optional[i], parent[i]});
  }
  TableSchema schema;
  buildSchema(nodes, -1, &schema);
  auto rw = new RefWriter();
  rw->w = CSTableWriter::createFile(path, schema);
  return rw;
  REF_CATCH(nullptr)
}

/* bulk append; rl/dl may be NULL (=> 0 / dlevel_max); present may be NULL (=> all present) */
int ref_writer_put_uint(
    void* h, const char* col, uint64_t n,
    const uint64_t* rl, const uint64_t* dl, const uint8_t* present,
    const uint64_t* v) {
  REF_TRY
  auto cw = ((RefWriter*) h)->w->getColumnWriter(col);
  auto dmax = cw->maxDefinitionLevel();
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t r = rl ? rl[i] : 0;
    uint64_t d = dl ? dl[i] : dmax;
    if (present && !present[i]) {
      if (!dl) d = dmax > 0 ? dmax - 1 : 0;
      cw->writeNull(r, d);
    } else if (d != dmax) {
      cw->writeNull(r, d);
    } else {
      cw->writeUnsignedInt(r, d, v[i]);
    }
  }
  return 0;
  REF_CATCH(-1)
}

int ref_writer_put_float(
    void* h, const char* col, uint64_t n,
    const uint64_t* rl, const uint64_t* dl, const uint8_t* present,
    const double* v) {
  REF_TRY
  auto cw = ((RefWriter*) h)->w->getColumnWriter(col);
  auto dmax = cw->maxDefinitionLevel();
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t r = rl ? rl[i] : 0;
    uint64_t d = dl ? dl[i] : dmax;
    if (present && !present[i]) {
      if (!dl) d = dmax > 0 ? dmax - 1 : 0;
      cw->writeNull(r, d);
    } else if (d != dmax) {
      cw->writeNull(r, d);
    } else {
      cw->writeFloat(r, d, v[i]);
    }
  }
  return 0;
  REF_CATCH(-1)
}

/* strings: offsets[n+1] into bytes */
int ref_writer_put_string(
    void* h, const char* col, uint64_t n,
    const uint64_t* rl, const uint64_t* dl, const uint8_t* present,
    const uint64_t* offsets, const char* bytes) {
  REF_TRY
  auto cw = ((RefWriter*) h)->w->getColumnWriter(col);
  auto dmax = cw->maxDefinitionLevel();
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t r = rl ? rl[i] : 0;
    uint64_t d = dl ? dl[i] : dmax;
    if (present && !present[i]) {
      if (!dl) d = dmax > 0 ? dmax - 1 : 0;
      cw->writeNull(r, d);
    } else if (d != dmax) {
      cw->writeNull(r, d);
    } else {
      cw->writeString(r, d, bytes + offsets[i], offsets[i + 1] - offsets[i]);
    }
  }
  return 0;
  REF_CATCH(-1)
}

/* Record by record, every column per record in the order given -- the call order of a
 * row-wise writer (cstable_writer.cc addRow loops; RecordShredder.cc:113-176 writes the
 * fields of one record before the next).  Column c has nslots[c] (r, d, value) triples;
 * a record takes the slots up to the next r == 0 (rl[c] == NULL: one slot per record).
 * kinds: 0 uint, 1 float.  dl[c] == NULL: present[c] (or all present) decides. */
int ref_writer_put_records(
    void* h, int ncols, const char* const* cols, const int* kinds,
    const uint64_t* nslots, const uint64_t* const* rl, const uint64_t* const* dl,
    const uint8_t* const* present, const uint64_t* const* values, uint64_t nrecords) {
  REF_TRY
  auto w = ((RefWriter*) h)->w;
  std::vector<RefPtr<ColumnWriter>> cw;
  std::vector<uint64_t> cur(ncols, 0);
  for (int c = 0; c < ncols; ++c) cw.push_back(w->getColumnWriter(cols[c]));
  for (uint64_t rec = 0; rec < nrecords; ++rec) {
    for (int c = 0; c < ncols; ++c) {
      auto dmax = cw[c]->maxDefinitionLevel();
      bool first = true;
      while (cur[c] < nslots[c]) {
        uint64_t i = cur[c];
        uint64_t r = rl[c] ? rl[c][i] : 0;
        if (!first && r == 0) break;
        if (!rl[c] && !first) break;
        first = false;
        uint64_t d = dl[c] ? dl[c][i] : dmax;
        if (!dl[c] && present[c] && !present[c][i]) d = dmax > 0 ? dmax - 1 : 0;
        if (d != dmax) {
          cw[c]->writeNull(r, d);
        } else if (kinds[c] == 1) {
          double f;
          memcpy(&f, &values[c][i], 8);
          cw[c]->writeFloat(r, d, f);
        } else {
          cw[c]->writeUnsignedInt(r, d, values[c][i]);
        }
        ++cur[c];
      }
    }
  }
  return 0;
  REF_CATCH(-1)
}

int ref_writer_commit(void* h, uint64_t nrows) {
  REF_TRY
  auto w = ((RefWriter*) h)->w;
  w->addRows(nrows);
  w->commit();
  return 0;
  REF_CATCH(-1)
}

void ref_writer_free(void* h) { delete (RefWriter*) h; }

void* ref_reader_open(const char* path) {
  REF_TRY
  auto rr = new RefReader();
  rr->r = CSTableReader::openFile(path);
  return rr;
  REF_CATCH(nullptr)
}

void ref_reader_free(void* h) { delete (RefReader*) h; }

uint64_t ref_reader_num_records(void* h) {
  return ((RefReader*) h)->r->numRecords();
}

int ref_reader_num_columns(void* h) {
  return (int) ((RefReader*) h)->r->columns().size();
}

/* name_out must hold >= 256 bytes */
int ref_reader_column_info(
    void* h, int idx, char* name_out,
    int* logical_type, int* storage_type, uint64_t* column_id,
    uint64_t* rlevel_max, uint64_t* dlevel_max) {
  REF_TRY
  const auto& c = ((RefReader*) h)->r->columns().at(idx);
  strncpy(name_out, c.column_name.c_str(), 255);
  name_out[255] = 0;
  *logical_type = (int) c.logical_type;
  *storage_type = (int) c.storage_type;
  *column_id = c.column_id;
  *rlevel_max = c.rlevel_max;
  *dlevel_max = c.dlevel_max;
  return 0;
  REF_CATCH(-1)
}

/* read the next n value slots of a column via a PRIVATE reader created per call
 * sequence: `cursor` is an opaque per-column reader handle. */
void* ref_column_open(void* h, const char* col) {
  REF_TRY
  auto rd = ((RefReader*) h)->r->getColumnReader(
      col, ColumnReader::Visibility::PRIVATE);
  rd->incRef();
  return rd.get();
  REF_CATCH(nullptr)
}

void ref_column_close(void* c) { ((ColumnReader*) c)->decRef(); }

int ref_column_read_uint(
    void* c, uint64_t n,
    uint64_t* rl, uint64_t* dl, uint8_t* present, uint64_t* v) {
  REF_TRY
  auto rd = (ColumnReader*) c;
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t r, d, val;
    bool p = rd->readUnsignedInt(&r, &d, &val);
    if (rl) rl[i] = r;
    if (dl) dl[i] = d;
    if (present) present[i] = p;
    v[i] = val;
  }
  return 0;
  REF_CATCH(-1)
}

int ref_column_read_float(
    void* c, uint64_t n,
    uint64_t* rl, uint64_t* dl, uint8_t* present, double* v) {
  REF_TRY
  auto rd = (ColumnReader*) c;
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t r, d;
    double val;
    bool p = rd->readFloat(&r, &d, &val);
    if (rl) rl[i] = r;
    if (dl) dl[i] = d;
    if (present) present[i] = p;
    v[i] = val;
  }
  return 0;
  REF_CATCH(-1)
}

/* strings are returned concatenated into `bytes` (capacity cap); offsets[n+1].
 * returns -2 if cap is too small. */
int ref_column_read_string(
    void* c, uint64_t n,
    uint64_t* rl, uint64_t* dl, uint8_t* present,
    uint64_t* offsets, char* bytes, uint64_t cap) {
  REF_TRY
  auto rd = (ColumnReader*) c;
  uint64_t pos = 0;
  offsets[0] = 0;
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t r, d;
    std::string val;
    bool p = rd->readString(&r, &d, &val);
    if (rl) rl[i] = r;
    if (dl) dl[i] = d;
    if (present) present[i] = p;
    if (pos + val.size() > cap) return -2;
    memcpy(bytes + pos, val.data(), val.size());
    pos += val.size();
    offsets[i + 1] = pos;
  }
  return 0;
  REF_CATCH(-1)
}

/* the reference SHA1 (src/eventql/util/SHA1.cc) -- pins the oracle's sha1 */
void ref_sha1(const void* data, uint64_t len, uint8_t out[20]) {
  auto h = SHA1::compute(data, len);
  memcpy(out, h.data(), 20);
}

}  // extern "C"
