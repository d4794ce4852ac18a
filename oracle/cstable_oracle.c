/*
 * cstable_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's cstable *read* path, value at a time, in
 * the reference's own order:
 *   container     io/cstable/cstable.cc:35-132 (v0.1.0), :200-255 (v0.2.0)
 *   page lookup   io/cstable/page_manager.cc:155-170
 *   page readers  io/cstable/columns/page_reader_{uint64,uint32,bitpacked,
 *                 leb128,ieee754,lenencstring}.cc
 *   column reader io/cstable/columns/column_reader_{uint,float,string}.cc
 *   v0.1.0        io/cstable/columns/v1/ColumnReader.h:36-52,
 *                 util/util/BitPackDecoder.{h,cc}
 *   bit layout    deps/3rdparty/libsimdcomp/simdbitpacking.c (4-lane vertical)
 */
#include "oracle.h"
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

static __thread char g_err[512];
const char* orc_last_error(void) { return g_err; }
static void set_err(const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
}

typedef struct {
  uint64_t offset;
  uint32_t size;
} pageref_t;

typedef struct {
  pageref_t* p;
  uint32_t n, cap;
} pagelist_t;

typedef struct {
  char name[256];
  int logical_type;
  int storage_type;
  uint64_t column_id;
  uint32_t rmax, dmax;
  /* v2 */
  pagelist_t data, rlevel, dlevel;
  /* v1 */
  uint64_t body_offset, body_size;
} colinfo_t;

struct orc_table {
  uint8_t* image;
  size_t len;
  int mapped; /* 1 = mmap, 0 = malloc copy */
  int version;
  uint64_t num_rows;
  int ncols;
  colinfo_t* cols;
};

/* ---- little helpers ----------------------------------------------------- */
static uint64_t rd_fixed(const uint8_t* p, int n) {
  uint64_t v = 0;
  for (int i = 0; i < n; ++i) v |= (uint64_t) p[i] << (8 * i);
  return v;
}

static int rd_varuint(const uint8_t* p, size_t len, size_t* pos, uint64_t* out) {
  uint64_t v = 0;
  for (int i = 0; i < 10; ++i) {
    if (*pos >= len) return -1;
    uint8_t b = p[(*pos)++];
    v |= (uint64_t) (b & 0x7f) << (7 * i);
    if (!(b & 0x80)) {
      *out = v;
      return 0;
    }
  }
  return -1;
}

static void pagelist_push(pagelist_t* l, pageref_t r) {
  if (l->n == l->cap) {
    l->cap = l->cap ? l->cap * 2 : 8;
    l->p = (pageref_t*) realloc(l->p, l->cap * sizeof(pageref_t));
  }
  l->p[l->n++] = r;
}

/* libsimdcomp bits() -- simdcomputil.c:9-20 */
static uint32_t bits_of(uint32_t v) {
  uint32_t b = 0;
  while (v) {
    ++b;
    v >>= 1;
  }
  return b;
}

/* simdunpack: 128 values of b bits, 4-lane vertical layout.
 * value i: lane l = i & 3, k = i >> 2; bit position p = k*b in the lane's
 * stream; lane word w lives at u32 index 4*w + l. */
static void unpack128(const uint8_t* in, uint32_t b, uint32_t* out) {
  uint32_t W[132];
  memset(W, 0, sizeof(W));
  memcpy(W, in, 16 * b);
  uint64_t mask = b >= 32 ? 0xffffffffull : ((1ull << b) - 1);
  for (uint32_t i = 0; i < 128; ++i) {
    uint32_t l = i & 3, k = i >> 2, p = k * b, w = p >> 5, s = p & 31;
    uint64_t v = (uint64_t) W[4 * w + l] >> s;
    if (s + b > 32) v |= (uint64_t) W[4 * (w + 1) + l] << (32 - s);
    out[i] = (uint32_t) (v & mask);
  }
}

/* ---- container ----------------------------------------------------------- */
static int parse_v1(orc_table_t* t) {
  /* cstable.cc:89-132 */
  const uint8_t* p = t->image;
  size_t pos = 6;
  pos += 8; /* flags */
  t->num_rows = rd_fixed(p + pos, 8);
  pos += 8;
  uint32_t ncols = (uint32_t) rd_fixed(p + pos, 4);
  pos += 4;
  t->ncols = (int) ncols;
  t->cols = (colinfo_t*) calloc(ncols, sizeof(colinfo_t));
  for (uint32_t i = 0; i < ncols; ++i) {
    colinfo_t* c = &t->cols[i];
    c->storage_type = (int) rd_fixed(p + pos, 4);
    pos += 4;
    switch (c->storage_type) {
      case EVQL_ENC_BOOLEAN_BITPACKED:
        c->logical_type = EVQL_COL_BOOLEAN;
        break;
      case EVQL_ENC_FLOAT_IEEE754:
        c->logical_type = EVQL_COL_FLOAT;
        break;
      case EVQL_ENC_STRING_PLAIN:
        c->logical_type = EVQL_COL_STRING;
        break;
      default:
        c->logical_type = EVQL_COL_UNSIGNED_INT;
    }
    uint32_t nl = (uint32_t) rd_fixed(p + pos, 4);
    pos += 4;
    if (nl > 255 || pos + nl > t->len) return -1;
    memcpy(c->name, p + pos, nl);
    c->name[nl] = 0;
    pos += nl;
    c->rmax = (uint32_t) rd_fixed(p + pos, 4);
    pos += 4;
    c->dmax = (uint32_t) rd_fixed(p + pos, 4);
    pos += 4;
    c->body_offset = rd_fixed(p + pos, 8);
    pos += 8;
    c->body_size = rd_fixed(p + pos, 8);
    pos += 8;
    if (c->body_offset + c->body_size > t->len) return -1;
  }
  return 0;
}

static int parse_v2(orc_table_t* t) {
  /* cstable.cc:152-171 (metablock), :200-227 (header), :245-255 (index) */
  const uint8_t* p = t->image;
  int have = 0;
  uint64_t txid = 0, index_off = 0;
  uint32_t index_size = 0;
  for (int i = 0; i < 2; ++i) {
    const uint8_t* mb = p + 14 + 48 * i;
    uint8_t h[20];
    orc_sha1(mb, 28, h);
    if (memcmp(h, mb + 28, 20) != 0) continue;
    uint64_t tx = rd_fixed(mb, 8);
    /* cstable.cc:69-75: strictly-greater picks block 0, else block 1 */
    if (!have || tx >= txid) {
      txid = tx;
      t->num_rows = rd_fixed(mb + 8, 8);
      index_off = rd_fixed(mb + 16, 8);
      index_size = (uint32_t) rd_fixed(mb + 24, 4);
      have = 1;
    }
  }
  if (!have) {
    set_err("can't open cstable: no valid metablocks found");
    return -1;
  }
  size_t pos = 14 + 96 + 128;
  uint64_t ncols;
  if (rd_varuint(p, t->len, &pos, &ncols)) return -1;
  t->ncols = (int) ncols;
  t->cols = (colinfo_t*) calloc(ncols ? ncols : 1, sizeof(colinfo_t));
  for (uint64_t i = 0; i < ncols; ++i) {
    colinfo_t* c = &t->cols[i];
    uint64_t v, nl;
    if (rd_varuint(p, t->len, &pos, &v)) return -1;
    c->logical_type = (int) v;
    if (rd_varuint(p, t->len, &pos, &v)) return -1;
    c->storage_type = (int) v;
    if (rd_varuint(p, t->len, &pos, &c->column_id)) return -1;
    if (rd_varuint(p, t->len, &pos, &nl)) return -1;
    if (nl > 255 || pos + nl > t->len) return -1;
    memcpy(c->name, p + pos, nl);
    c->name[nl] = 0;
    pos += nl;
    if (rd_varuint(p, t->len, &pos, &v)) return -1;
    c->rmax = (uint32_t) v;
    if (rd_varuint(p, t->len, &pos, &v)) return -1;
    c->dmax = (uint32_t) v;
  }
  if (index_off + index_size > t->len) return -1;
  size_t ipos = index_off, iend = index_off + index_size;
  uint64_t n;
  if (rd_varuint(p, iend, &ipos, &n)) return -1;
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t kind, cid, off, size;
    if (rd_varuint(p, iend, &ipos, &kind)) return -1;
    if (rd_varuint(p, iend, &ipos, &cid)) return -1;
    if (rd_varuint(p, iend, &ipos, &off)) return -1;
    if (rd_varuint(p, iend, &ipos, &size)) return -1;
    if (off + size > t->len) return -1;
    /* PageManager::getPages: index order, matching (column_id, entry_type) */
    for (int c = 0; c < t->ncols; ++c) {
      if (t->cols[c].column_id != cid) continue;
      pageref_t r = {off, (uint32_t) size};
      if (kind == 1) pagelist_push(&t->cols[c].data, r);
      if (kind == 2) pagelist_push(&t->cols[c].rlevel, r);
      if (kind == 3) pagelist_push(&t->cols[c].dlevel, r);
    }
  }
  return 0;
}

static orc_table_t* table_from(uint8_t* image, size_t len, int mapped) {
  static const uint8_t magic[4] = {0x23, 0x17, 0x23, 0x17};
  orc_table_t* t = (orc_table_t*) calloc(1, sizeof(orc_table_t));
  t->image = image;
  t->len = len;
  t->mapped = mapped;
  int rc = -1;
  if (len >= 26 && memcmp(image, magic, 4) == 0) {
    t->version = image[4];
    if (t->version == 1) rc = parse_v1(t);
    else if (t->version == 2 && len >= 512) rc = parse_v2(t);
    else set_err("unsupported cstable version");
  } else {
    set_err("not a valid cstable file");
  }
  if (rc != 0) {
    if (!g_err[0]) set_err("corrupt cstable file");
    orc_table_close(t);
    return NULL;
  }
  return t;
}

orc_table_t* orc_table_open(const char* path) {
  g_err[0] = 0;
  int fd = open(path, O_RDONLY);
  if (fd < 0) {
    set_err("can't open file");
    return NULL;
  }
  struct stat st;
  fstat(fd, &st);
  void* m = mmap(NULL, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) {
    set_err("mmap failed");
    return NULL;
  }
  return table_from((uint8_t*) m, st.st_size, 1);
}

orc_table_t* orc_table_open_image(const void* image, size_t len) {
  g_err[0] = 0;
  uint8_t* cp = (uint8_t*) malloc(len ? len : 1);
  memcpy(cp, image, len);
  return table_from(cp, len, 0);
}

void orc_table_close(orc_table_t* t) {
  if (!t) return;
  for (int i = 0; i < t->ncols; ++i) {
    free(t->cols[i].data.p);
    free(t->cols[i].rlevel.p);
    free(t->cols[i].dlevel.p);
  }
  free(t->cols);
  if (t->mapped) munmap(t->image, t->len);
  else free(t->image);
  free(t);
}

int orc_table_version(const orc_table_t* t) { return t->version; }
uint64_t orc_table_num_rows(const orc_table_t* t) { return t->num_rows; }
int orc_table_num_columns(const orc_table_t* t) { return t->ncols; }

int orc_table_column_info(const orc_table_t* t, int idx, char* name_out,
                          int* logical_type, int* storage_type,
                          uint64_t* column_id, uint64_t* rlevel_max,
                          uint64_t* dlevel_max) {
  if (idx < 0 || idx >= t->ncols) return -1;
  const colinfo_t* c = &t->cols[idx];
  strcpy(name_out, c->name);
  *logical_type = c->logical_type;
  *storage_type = c->storage_type;
  *column_id = c->column_id;
  *rlevel_max = c->rmax;
  *dlevel_max = c->dmax;
  return 0;
}

static const colinfo_t* find_col(const orc_table_t* t, const char* name) {
  for (int i = 0; i < t->ncols; ++i) {
    if (strcmp(t->cols[i].name, name) == 0) return &t->cols[i];
  }
  return NULL;
}

uint64_t orc_table_column_num_values(const orc_table_t* t, const char* name) {
  const colinfo_t* c = find_col(t, name);
  if (!c || t->version != 1) return 0;
  return rd_fixed(t->image + c->body_offset, 8);
}

/* ---- page streams ---------------------------------------------------------
 * A byte stream over a list of pages (v2) or one contiguous region (v1). */
typedef struct {
  const uint8_t* base;
  const pageref_t* pages;
  uint32_t npages;
  uint32_t page_idx; /* next page to load */
  const uint8_t* page_data;
  uint64_t page_pos, page_len;
  pageref_t single;
  int is_region; /* pages == &single (must be re-pointed after a struct copy) */
} pstream_t;

static void ps_fix(pstream_t* s) {
  if (s->is_region) s->pages = &s->single;
}

static void ps_init_pages(pstream_t* s, const orc_table_t* t,
                          const pagelist_t* l) {
  memset(s, 0, sizeof(*s));
  s->base = t->image;
  s->pages = l->p;
  s->npages = l->n;
}

static void ps_init_region(pstream_t* s, const orc_table_t* t, uint64_t off,
                           uint64_t size) {
  memset(s, 0, sizeof(*s));
  s->base = t->image;
  s->single.offset = off;
  s->single.size = (uint32_t) size;
  s->is_region = 1;
  s->pages = &s->single;
  s->npages = size > 0 ? 1 : 0;
}

static int ps_next_page(pstream_t* s) {
  if (s->page_idx == s->npages) return 0;
  s->page_pos = 0;
  s->page_len = s->pages[s->page_idx].size;
  s->page_data = s->base + s->pages[s->page_idx].offset;
  ++s->page_idx;
  return 1;
}

/* ---- unsigned int page readers (UnsignedIntPageReader) --------------------
 * all keep the reference's one-value-lookahead (cur_val / eof) behaviour */
enum { RD_NONE = 0, RD_U64, RD_U32, RD_LEB128, RD_BITPACKED };

typedef struct {
  int kind;
  pstream_t ps;
  uint64_t cur_val;
  int eof;
  /* bitpacked */
  uint32_t maxbits;
  uint32_t outbuf[128];
  uint32_t outbuf_pos;
} uintreader_t;

static void ur_fetch_next(uintreader_t* r);

/* page_reader_bitpacked.cc:91-111 */
static void ur_fetch_batch(uintreader_t* r) {
  uint32_t batch_size = 16 * r->maxbits;
  uint8_t batch[512];
  memset(batch, 0, sizeof(batch));
  for (uint32_t b = 0; b < batch_size;) {
    if (r->ps.page_pos == r->ps.page_len) {
      if (!ps_next_page(&r->ps)) break;
    }
    uint32_t c = (uint32_t) (r->ps.page_len - r->ps.page_pos);
    if (c > batch_size - b) c = batch_size - b;
    memcpy(batch + b, r->ps.page_data + r->ps.page_pos, c);
    r->ps.page_pos += c;
    b += c;
  }
  unpack128(batch, r->maxbits, r->outbuf);
  r->outbuf_pos = 0;
}

static void ur_fetch_next(uintreader_t* r) {
  switch (r->kind) {
    case RD_U64: /* page_reader_uint64.cc:50-70 */
    case RD_U32: {
      uint32_t w = r->kind == RD_U64 ? 8 : 4;
      if (r->ps.page_pos + w > r->ps.page_len) {
        if (!ps_next_page(&r->ps)) {
          r->eof = 1;
          return;
        }
      }
      r->cur_val = rd_fixed(r->ps.page_data + r->ps.page_pos, w);
      r->ps.page_pos += w;
      return;
    }
    case RD_LEB128: /* page_reader_leb128.cc:50-73 */
      r->cur_val = 0;
      for (int i = 0;; ++i) {
        if (r->ps.page_pos >= r->ps.page_len) {
          if (!ps_next_page(&r->ps)) {
            r->eof = 1;
            return;
          }
        }
        uint8_t b = r->ps.page_data[r->ps.page_pos++];
        r->cur_val |= (uint64_t) (b & 0x7f) << (7 * i);
        if (!(b & 0x80)) break;
      }
      return;
    case RD_BITPACKED: /* page_reader_bitpacked.cc:60-76 */
      if (r->eof) return;
      if (r->maxbits == 0) {
        r->cur_val = 0;
        return;
      }
      if (r->outbuf_pos == 128) ur_fetch_batch(r);
      r->cur_val = r->outbuf[r->outbuf_pos++];
      return;
  }
}

/* v2 constructors */
static void ur_open_plain(uintreader_t* r, int kind, const orc_table_t* t,
                          const pagelist_t* l) {
  memset(r, 0, sizeof(*r));
  r->kind = kind;
  ps_init_pages(&r->ps, t, l);
  ur_fetch_next(r);
}

/* page_reader_bitpacked.cc:30-48: max_value prefix on the first page */
static void ur_open_bitpacked_prefixed(uintreader_t* r, pstream_t ps) {
  memset(r, 0, sizeof(*r));
  r->kind = RD_BITPACKED;
  r->ps = ps;
  ps_fix(&r->ps);
  r->outbuf_pos = 128;
  if (r->ps.npages > 0) {
    ps_next_page(&r->ps);
    uint32_t max_val = (uint32_t) rd_fixed(r->ps.page_data, 4);
    r->ps.page_pos = 4;
    r->maxbits = max_val > 0 ? bits_of(max_val) : 0;
    ur_fetch_next(r);
  } else {
    r->eof = 1;
  }
}

/* v1 util::BitPackDecoder: no prefix, width from the header's max value */
static void ur_open_bitpacked_raw(uintreader_t* r, pstream_t ps,
                                  uint32_t max_val) {
  memset(r, 0, sizeof(*r));
  r->kind = RD_BITPACKED;
  r->ps = ps;
  ps_fix(&r->ps);
  r->outbuf_pos = 128;
  r->maxbits = max_val > 0 ? bits_of(max_val) : 0;
  if (r->ps.npages > 0) ps_next_page(&r->ps);
  /* no lookahead here: BitPackDecoder::next/peek fetch lazily */
}

static uint64_t ur_read(uintreader_t* r) {
  uint64_t v = r->cur_val;
  ur_fetch_next(r);
  return v;
}

/* v1 decoder semantics (BitPackDecoder::next / ::peek) */
static uint32_t bpd_next(uintreader_t* r) {
  if (r->maxbits == 0) return 0;
  if (r->outbuf_pos == 128) ur_fetch_batch(r);
  return r->outbuf[r->outbuf_pos++];
}
static uint32_t bpd_peek(uintreader_t* r) {
  if (r->maxbits == 0) return 0;
  if (r->outbuf_pos == 128) ur_fetch_batch(r);
  return r->outbuf[r->outbuf_pos];
}

/* ---- column cursor -------------------------------------------------------- */
struct orc_column {
  const orc_table_t* t;
  const colinfo_t* info;
  int version;
  uintreader_t rl, dl;
  int has_rl, has_dl;
  /* data */
  uintreader_t udata; /* uint encodings */
  pstream_t bytes;    /* ieee754 / strings (v2), raw data region (v1) */
  int data_kind;      /* storage_type */
};

orc_column_t* orc_column_open(orc_table_t* t, const char* name) {
  const colinfo_t* ci = find_col(t, name);
  if (!ci) {
    set_err("column not found");
    return NULL;
  }
  orc_column_t* c = (orc_column_t*) calloc(1, sizeof(orc_column_t));
  c->t = t;
  c->info = ci;
  c->version = t->version;
  c->data_kind = ci->storage_type;
  if (t->version == 2) {
    /* cstable_reader.cc:81-131 openColumnV2 */
    if (ci->rmax > 0) {
      pstream_t ps;
      ps_init_pages(&ps, t, &ci->rlevel);
      ur_open_bitpacked_prefixed(&c->rl, ps);
      c->has_rl = 1;
    }
    if (ci->dmax > 0) {
      pstream_t ps;
      ps_init_pages(&ps, t, &ci->dlevel);
      ur_open_bitpacked_prefixed(&c->dl, ps);
      c->has_dl = 1;
    }
    switch (ci->storage_type) {
      case EVQL_ENC_UINT64_PLAIN:
        ur_open_plain(&c->udata, RD_U64, t, &ci->data);
        break;
      case EVQL_ENC_UINT32_PLAIN:
        ur_open_plain(&c->udata, RD_U32, t, &ci->data);
        break;
      case EVQL_ENC_UINT64_LEB128:
        ur_open_plain(&c->udata, RD_LEB128, t, &ci->data);
        break;
      case EVQL_ENC_UINT32_BITPACKED:
      case EVQL_ENC_BOOLEAN_BITPACKED: {
        pstream_t ps;
        ps_init_pages(&ps, t, &ci->data);
        ur_open_bitpacked_prefixed(&c->udata, ps);
        break;
      }
      default:
        ps_init_pages(&c->bytes, t, &ci->data);
    }
  } else {
    /* v1/ColumnReader.h:36-52 */
    const uint8_t* body = t->image + ci->body_offset;
    uint64_t rs = rd_fixed(body + 8, 8), ds = rd_fixed(body + 16, 8);
    uint64_t dsz = rd_fixed(body + 24, 8);
    uint64_t roff = ci->body_offset + 32, doff = roff + rs, voff = doff + ds;
    pstream_t ps;
    ps_init_region(&ps, t, roff, rs);
    ur_open_bitpacked_raw(&c->rl, ps, ci->rmax);
    ps_init_region(&ps, t, doff, ds);
    ur_open_bitpacked_raw(&c->dl, ps, ci->dmax);
    c->has_rl = c->has_dl = 1;
    switch (ci->storage_type) {
      case EVQL_ENC_UINT32_BITPACKED: {
        /* v1/BitPackedIntColumnReader.cc:31-41: u32 max then blocks */
        uint32_t maxv = (uint32_t) rd_fixed(t->image + voff, 4);
        ps_init_region(&ps, t, voff + 4, dsz >= 4 ? dsz - 4 : 0);
        ur_open_bitpacked_raw(&c->udata, ps, maxv);
        break;
      }
      case EVQL_ENC_BOOLEAN_BITPACKED:
        ps_init_region(&ps, t, voff, dsz);
        ur_open_bitpacked_raw(&c->udata, ps, 1);
        break;
      default:
        ps_init_region(&c->bytes, t, voff, dsz);
        if (c->bytes.npages) ps_next_page(&c->bytes);
    }
  }
  return c;
}

void orc_column_close(orc_column_t* c) { free(c); }

uint64_t orc_column_next_rlevel(orc_column_t* c) {
  if (c->version == 2) {
    /* DefaultColumnReader::nextRepetitionLevel, ColumnReader.cc:58-64 */
    return c->info->rmax > 0 ? c->rl.cur_val : 0;
  }
  return bpd_peek(&c->rl);
}

static void read_levels(orc_column_t* c, uint64_t* r, uint64_t* d) {
  if (c->version == 2) {
    *r = c->has_rl ? ur_read(&c->rl) : 0;
    *d = c->has_dl ? ur_read(&c->dl) : 0;
  } else {
    *r = bpd_next(&c->rl);
    *d = bpd_next(&c->dl);
  }
}

/* raw fixed-width / leb128 reads from the byte region (v1 data, v2 ieee754) */
static uint64_t bytes_fixed(pstream_t* s, int w) {
  /* page_reader_ieee754.cc:38-59: returns 0 when exhausted */
  if (s->page_pos + w > s->page_len) {
    if (!ps_next_page(s)) return 0;
  }
  uint64_t v = rd_fixed(s->page_data + s->page_pos, w);
  s->page_pos += w;
  return v;
}

static int bytes_byte(pstream_t* s, uint8_t* out) {
  if (s->page_pos >= s->page_len) {
    if (!ps_next_page(s)) return -1;
  }
  *out = s->page_data[s->page_pos++];
  return 0;
}

static uint64_t bytes_leb128(pstream_t* s) {
  uint64_t v = 0;
  for (int i = 0;; ++i) {
    uint8_t b;
    if (bytes_byte(s, &b)) break;
    v |= (uint64_t) (b & 0x7f) << (7 * i);
    if (!(b & 0x80)) break;
  }
  return v;
}

static uint64_t read_uint_value(orc_column_t* c) {
  if (c->version == 2) {
    switch (c->data_kind) {
      case EVQL_ENC_UINT64_PLAIN:
      case EVQL_ENC_UINT32_PLAIN:
      case EVQL_ENC_UINT64_LEB128:
      case EVQL_ENC_UINT32_BITPACKED:
      case EVQL_ENC_BOOLEAN_BITPACKED:
        return ur_read(&c->udata);
      case EVQL_ENC_FLOAT_IEEE754: {
        /* FloatColumnReader::readUnsignedInt casts the double */
        uint64_t bits = bytes_fixed(&c->bytes, 8);
        double dv;
        memcpy(&dv, &bits, 8);
        return (uint64_t) dv;
      }
    }
    return 0;
  }
  switch (c->data_kind) {
    case EVQL_ENC_UINT32_BITPACKED:
    case EVQL_ENC_BOOLEAN_BITPACKED:
      return bpd_next(&c->udata);
    case EVQL_ENC_UINT32_PLAIN:
      return bytes_fixed(&c->bytes, 4);
    case EVQL_ENC_UINT64_PLAIN:
      return bytes_fixed(&c->bytes, 8);
    case EVQL_ENC_UINT64_LEB128:
      return bytes_leb128(&c->bytes);
    case EVQL_ENC_FLOAT_IEEE754: {
      uint64_t bits = bytes_fixed(&c->bytes, 8);
      double dv;
      memcpy(&dv, &bits, 8);
      return (uint64_t) dv;
    }
  }
  return 0;
}

/* UnsignedIntColumnReader::readUnsignedInt, column_reader_uint.cc:92-115 */
int orc_column_read_uint(orc_column_t* c, uint64_t n, uint64_t* rl,
                         uint64_t* dl, uint8_t* present, uint64_t* v) {
  if (c->data_kind == EVQL_ENC_STRING_PLAIN) {
    set_err("read_uint on string column");
    return -1;
  }
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t r, d;
    read_levels(c, &r, &d);
    int p = (d == c->info->dmax);
    if (rl) rl[i] = r;
    if (dl) dl[i] = d;
    if (present) present[i] = (uint8_t) p;
    v[i] = p ? read_uint_value(c) : 0;
  }
  return 0;
}

/* FloatColumnReader::readFloat, column_reader_float.cc:103-126; on a uint
 * column UnsignedIntColumnReader::readFloat casts (column_reader_uint.cc) */
int orc_column_read_float(orc_column_t* c, uint64_t n, uint64_t* rl,
                          uint64_t* dl, uint8_t* present, double* v) {
  if (c->data_kind == EVQL_ENC_STRING_PLAIN) {
    set_err("read_float on string column");
    return -1;
  }
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t r, d;
    read_levels(c, &r, &d);
    int p = (d == c->info->dmax);
    if (rl) rl[i] = r;
    if (dl) dl[i] = d;
    if (present) present[i] = (uint8_t) p;
    if (!p) {
      v[i] = 0;
    } else if (c->data_kind == EVQL_ENC_FLOAT_IEEE754) {
      uint64_t bits = bytes_fixed(&c->bytes, 8);
      memcpy(&v[i], &bits, 8);
    } else {
      v[i] = (double) read_uint_value(c);
    }
  }
  return 0;
}

/* StringColumnReader::readString, column_reader_string.cc:121-145 +
 * LenencStringPageReader::readString, page_reader_lenencstring.cc:37-62 */
int orc_column_read_string(orc_column_t* c, uint64_t n, uint64_t* rl,
                           uint64_t* dl, uint8_t* present, uint64_t* offsets,
                           char* bytes, uint64_t cap) {
  if (c->data_kind != EVQL_ENC_STRING_PLAIN) {
    set_err("read_string on non-string column");
    return -1;
  }
  uint64_t pos = 0;
  offsets[0] = 0;
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t r, d;
    read_levels(c, &r, &d);
    int p = (d == c->info->dmax);
    if (rl) rl[i] = r;
    if (dl) dl[i] = d;
    if (present) present[i] = (uint8_t) p;
    if (p) {
      uint64_t len;
      if (c->version == 2) {
        len = bytes_leb128(&c->bytes);
      } else {
        len = bytes_fixed(&c->bytes, 4); /* v1/StringColumnReader: u32 len */
      }
      if (pos + len > cap) return -2;
      for (uint64_t k = 0; k < len; ++k) {
        uint8_t b;
        if (bytes_byte(&c->bytes, &b)) {
          set_err("end of column reached");
          return -1;
        }
        bytes[pos++] = (char) b;
      }
    }
    offsets[i + 1] = pos;
  }
  return 0;
}

/* one string value into a growable buffer (same reads as orc_column_read_string) */
int orc_column_read_string_alloc(orc_column_t* c, uint64_t* r, uint64_t* d,
                                 uint8_t* present, char** buf, uint64_t* cap,
                                 uint64_t* len) {
  if (c->data_kind != EVQL_ENC_STRING_PLAIN) {
    set_err("read_string on non-string column");
    return -1;
  }
  read_levels(c, r, d);
  int p = (*d == c->info->dmax);
  *present = (uint8_t) p;
  *len = 0;
  if (!p) return 0;
  uint64_t l = c->version == 2 ? bytes_leb128(&c->bytes) : bytes_fixed(&c->bytes, 4);
  if (l > *cap) {
    *cap = l * 2;
    *buf = (char*) realloc(*buf, *cap);
  }
  for (uint64_t k = 0; k < l; ++k) {
    uint8_t b;
    if (bytes_byte(&c->bytes, &b)) {
      set_err("end of column reached");
      return -1;
    }
    (*buf)[k] = (char) b;
  }
  *len = l;
  return 0;
}

uint32_t orc_column_rmax(const orc_column_t* c) { return c->info->rmax; }
uint32_t orc_column_dmax(const orc_column_t* c) { return c->info->dmax; }
int orc_column_logical_type(const orc_column_t* c) { return c->info->logical_type; }
