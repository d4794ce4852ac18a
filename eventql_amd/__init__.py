"""eventql_amd -- MI355X-native scan -> filter -> GROUP BY executor for EventQL
cstable files, behind the C ABI of include/evql_gpu.h.

This package is a thin ctypes host layer over libevql_mi355x.so (C++/HIP).  It
never computes results itself: if the library, the HIP device or the kernel
compiler is missing, the calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import capi as K
from .plan import Plan, unpack_svector  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# EVQL_LIB: an instrumented build of the same library (host-side sanitizer runs)
LIB_PATH = os.environ.get("EVQL_LIB") or os.path.join(_HERE, "libevql_mi355x.so")
KERNEL_CACHE_DIR = os.path.join(_HERE, "_kcache")

_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)
_f64p = C.POINTER(C.c_double)


class EvqlError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (status %d)" % (msg, code))
        self.code = code
        self.msg = msg


_lib = None


def lib():
    """loads libevql_mi355x.so; raises if it has not been built"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libevql_mi355x.so is missing: run `make -C eventql_amd/csrc` "
            "(or __graft_entry__.build()); there is no fallback path")
    # PyTorch bundles its own HIP runtime (same soname, libamdhip64.so.7).  Two HIP
    # runtimes in one process cannot both own the GPU, so make sure torch's copy
    # is the one already loaded when our library's NEEDED entries are resolved.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.evql_last_error.restype = C.c_char_p
    L.evql_version.restype = C.c_char_p
    L.evql_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.evql_ctx_destroy.argtypes = [C.c_void_p]
    L.evql_ctx_synchronize.argtypes = [C.c_void_p]
    L.evql_ctx_stream.restype = C.c_void_p
    L.evql_ctx_stream.argtypes = [C.c_void_p]
    L.evql_table_open_file.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]
    L.evql_table_open_image.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.evql_table_close.argtypes = [C.c_void_p]
    L.evql_table_num_rows.restype = C.c_uint64
    L.evql_table_num_rows.argtypes = [C.c_void_p]
    L.evql_table_num_columns.argtypes = [C.c_void_p]
    L.evql_table_column_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(K.ColumnInfo)]
    L.evql_table_image_size.restype = C.c_uint64
    L.evql_table_image_size.argtypes = [C.c_void_p]
    L.evql_table_download_image.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.evql_table_generate.argtypes = [C.c_void_p, C.POINTER(K.SynthSpec), C.POINTER(C.c_void_p)]
    L.evql_table_from_device_columns.argtypes = [C.c_void_p, C.POINTER(K.ColumnSpec), C.c_int,
                                                 C.POINTER(K.DeviceColumn), C.c_uint64,
                                                 C.POINTER(C.c_void_p)]
    L.evql_table_from_device_columns_ordered.argtypes = [
        C.c_void_p, C.POINTER(K.ColumnSpec), C.c_int, C.POINTER(K.DeviceColumn), C.c_uint64,
        C.c_int, C.POINTER(C.c_void_p)]
    L.evql_writer_create.argtypes = [C.POINTER(K.ColumnSpec), C.c_int, C.POINTER(C.c_void_p)]
    L.evql_writer_put_uint.argtypes = [C.c_void_p, C.c_int, C.c_uint64, _u64p, _u64p, _u8p, _u64p]
    L.evql_writer_put_float.argtypes = [C.c_void_p, C.c_int, C.c_uint64, _u64p, _u64p, _u8p, _f64p]
    L.evql_writer_put_string.argtypes = [C.c_void_p, C.c_int, C.c_uint64, _u64p, _u64p, _u8p,
                                         _u64p, C.c_char_p]
    L.evql_writer_commit.argtypes = [C.c_void_p, C.c_uint64]
    L.evql_writer_image.restype = C.c_void_p
    L.evql_writer_image.argtypes = [C.c_void_p, _u64p]
    L.evql_cstable_upgrade.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, _u64p]
    L.evql_cstable_inspect.argtypes = [C.c_char_p, C.c_uint64, _u64p, C.POINTER(C.c_int)]
    L.evql_hub_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.evql_hub_destroy.argtypes = [C.c_void_p]
    L.evql_exchange_create_hub.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.evql_exchange_create_rccl.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p,
                                            C.POINTER(C.c_void_p)]
    L.evql_exchange_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(K.Transport),
                                       C.POINTER(C.c_void_p)]
    L.evql_rccl_unique_id.argtypes = [C.c_char_p]
    L.evql_exchange_destroy.argtypes = [C.c_void_p]
    L.evql_exchange_backend.restype = C.c_char_p
    L.evql_exchange_backend.argtypes = [C.c_void_p]
    L.evql_query_exchange.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.evql_exchange_last_stats.argtypes = [C.c_void_p, C.POINTER(K.ExchangeStats)]
    L.evql_writer_write_file.argtypes = [C.c_void_p, C.c_char_p]
    L.evql_writer_destroy.argtypes = [C.c_void_p]
    L.evql_query_create.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(K.PlanDesc),
                                    C.POINTER(C.c_void_p)]
    L.evql_query_destroy.argtypes = [C.c_void_p]
    L.evql_query_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.evql_query_launch.argtypes = [C.c_void_p]
    L.evql_query_finish.argtypes = [C.c_void_p]
    L.evql_query_column_count.argtypes = [C.c_void_p]
    L.evql_query_column_type.argtypes = [C.c_void_p, C.c_int]
    L.evql_query_next_batch.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(K.ColumnBuf),
                                        C.POINTER(C.c_size_t)]
    L.evql_query_stats.argtypes = [C.c_void_p, C.POINTER(K.QueryStats)]
    L.evql_query_kernel_source.restype = C.c_char_p
    L.evql_query_kernel_source.argtypes = [C.c_void_p]
    L.evql_query_record_words.restype = C.c_uint32
    L.evql_query_record_words.argtypes = [C.c_void_p]
    L.evql_query_partial_view.argtypes = [C.c_void_p, C.POINTER(K.PartialView)]
    L.evql_query_export_groups.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, _u64p]
    L.evql_query_import_groups.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.evql_query_reset.argtypes = [C.c_void_p]
    L.evql_query_distinct_aggregates.restype = C.c_uint32
    L.evql_query_distinct_aggregates.argtypes = [C.c_void_p]
    L.evql_query_export_pairs.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, _u64p]
    L.evql_query_import_pairs.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
    L.evql_query_set_order.argtypes = [C.c_void_p, C.POINTER(K.SortSpec), C.c_uint32, C.c_int64,
                                       C.c_uint64]
    L.evql_lsm_chain_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.evql_lsm_chain_destroy.argtypes = [C.c_void_p]
    L.evql_lsm_chain_add.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, _u8p, C.c_uint64]
    L.evql_lsm_chain_length.argtypes = [C.c_void_p]
    L.evql_query_create_chain.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(K.PlanDesc),
                                          C.POINTER(C.c_void_p)]
    L.evql_lsm_chain_build.argtypes = [C.c_void_p]
    L.evql_lsm_chain_filter.argtypes = [C.c_void_p, C.c_int, C.POINTER(_u8p), _u64p, _u64p]
    L.evql_merge_create.argtypes = [C.POINTER(K.PlanDesc), C.POINTER(C.c_void_p)]
    L.evql_merge_destroy.argtypes = [C.c_void_p]
    L.evql_merge_add_frame.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.evql_merge_add_rows.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p,
                                      C.c_size_t, C.c_size_t]
    L.evql_merge_num_groups.restype = C.c_uint64
    L.evql_merge_num_groups.argtypes = [C.c_void_p]
    L.evql_merge_column_count.restype = C.c_size_t
    L.evql_merge_column_count.argtypes = [C.c_void_p]
    L.evql_merge_column_type.restype = C.c_uint32
    L.evql_merge_column_type.argtypes = [C.c_void_p, C.c_size_t]
    L.evql_merge_next_batch.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(K.ColumnBuf),
                                        C.POINTER(C.c_size_t)]
    L.evql_set_kernel_cache_dir.argtypes = [C.c_char_p]
    L.evql_compile_only.argtypes = [C.POINTER(K.PlanDesc), C.POINTER(K.ColumnInfo), C.c_int,
                                    C.c_char_p, C.POINTER(C.c_size_t)]
    L.evql_set_kernel_cache_dir(KERNEL_CACHE_DIR.encode())
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise EvqlError(rc, lib().evql_last_error().decode(errors="replace"))


def _ptr(a, ty):
    return a.ctypes.data_as(ty) if a is not None else None


def inspect_image(image):
    """(num_rows, num_columns) of a cstable image, or EvqlError(EVQL_EIO) for a
    truncated / corrupt one (evql_cstable_inspect; host-only)"""
    buf = bytes(image)
    n = C.c_uint64(0)
    nc = C.c_int(0)
    _check(lib().evql_cstable_inspect(buf, len(buf), C.byref(n), C.byref(nc)))
    return n.value, nc.value


def upgrade_image(image):
    """cstable v0.1.0 image -> v0.2.0 image (evql_cstable_upgrade; host-only)"""
    buf = bytes(image)
    n = C.c_uint64()
    rc = lib().evql_cstable_upgrade(buf, len(buf), None, 0, C.byref(n))
    if rc != K.EVQL_EARG or n.value == 0:
        _check(rc)
    out = C.create_string_buffer(n.value)
    _check(lib().evql_cstable_upgrade(buf, len(buf), out, n.value, C.byref(n)))
    return out.raw[:n.value]


# ---------------------------------------------------------------------------
# host-side writer
# ---------------------------------------------------------------------------
class Writer:
    """cstable v0.2.0 writer (evql_writer_*).  columns: list of dict(name,
    logical_type, storage_type, rlevel_max=0, dlevel_max=0, bitpack_max_value=0)"""

    def __init__(self, columns):
        L = lib()
        self.columns = columns
        self._names = [c["name"].encode() for c in columns]
        specs = (K.ColumnSpec * len(columns))()
        for i, c in enumerate(columns):
            specs[i] = K.ColumnSpec(self._names[i], c["logical_type"], c["storage_type"],
                                    c.get("column_id", i + 1), c.get("rlevel_max", 0),
                                    c.get("dlevel_max", 0), c.get("bitpack_max_value", 0))
        self.h = C.c_void_p()
        _check(L.evql_writer_create(specs, len(columns), C.byref(self.h)))
        self._index = {c["name"]: i for i, c in enumerate(columns)}

    def put(self, name, values, rlvl=None, dlvl=None, present=None):
        L = lib()
        col = self._index[name]
        kind = self.columns[col]["logical_type"]
        n = len(values)
        rl = np.ascontiguousarray(rlvl, np.uint64) if rlvl is not None else None
        dl = np.ascontiguousarray(dlvl, np.uint64) if dlvl is not None else None
        pr = np.ascontiguousarray(present, np.uint8) if present is not None else None
        if kind == K.COL_STRING:
            off = np.zeros(n + 1, np.uint64)
            if n:
                off[1:] = np.cumsum([len(s) for s in values])
            blob = b"".join(values)
            _check(L.evql_writer_put_string(self.h, col, n, _ptr(rl, _u64p), _ptr(dl, _u64p),
                                            _ptr(pr, _u8p), _ptr(off, _u64p), blob))
        elif kind == K.COL_FLOAT:
            v = np.ascontiguousarray(values, np.float64)
            _check(L.evql_writer_put_float(self.h, col, n, _ptr(rl, _u64p), _ptr(dl, _u64p),
                                           _ptr(pr, _u8p), _ptr(v, _f64p)))
        else:
            v = np.ascontiguousarray(values, np.uint64)
            _check(L.evql_writer_put_uint(self.h, col, n, _ptr(rl, _u64p), _ptr(dl, _u64p),
                                          _ptr(pr, _u8p), _ptr(v, _u64p)))

    def commit(self, num_rows):
        _check(lib().evql_writer_commit(self.h, num_rows))

    def image(self):
        n = C.c_uint64()
        p = lib().evql_writer_image(self.h, C.byref(n))
        # (ctypes.string_at takes a C int: images beyond 2 GiB need the array form)
        return bytes((C.c_ubyte * n.value).from_address(p))

    def write_file(self, path):
        _check(lib().evql_writer_write_file(self.h, path.encode()))

    def close(self):
        if self.h:
            lib().evql_writer_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


# ---------------------------------------------------------------------------
# device objects
# ---------------------------------------------------------------------------
class Context:
    def __init__(self, device=0, stream=None):
        self.h = C.c_void_p()
        _check(lib().evql_ctx_create(device, stream, C.byref(self.h)))

    def synchronize(self):
        _check(lib().evql_ctx_synchronize(self.h))

    @property
    def stream(self):
        return lib().evql_ctx_stream(self.h)

    def open_file(self, path):
        t = C.c_void_p()
        _check(lib().evql_table_open_file(self.h, path.encode(), C.byref(t)))
        return Table(self, t)

    def open_image(self, image):
        t = C.c_void_p()
        buf = bytes(image)
        _check(lib().evql_table_open_image(self.h, buf, len(buf), C.byref(t)))
        return Table(self, t)

    def generate(self, num_rows, columns="kabv", seed=None, k_mod=1000, u_mod=0, k_bits=0):
        from . import synth
        bits = {"k": 1, "a": 2, "b": 4, "v": 8, "u": 16}
        mask = 0
        for c in columns:
            mask |= bits[c]
        spec = K.SynthSpec(num_rows, synth.SEED if seed is None else seed, k_mod, u_mod, mask,
                           k_bits)
        t = C.c_void_p()
        _check(lib().evql_table_generate(self.h, C.byref(spec), C.byref(t)))
        return Table(self, t)

    def table_from_device_columns(self, columns, values, nulls, num_rows, heaps=None,
                                  levels=None, page_order=0):
        """evql_table_from_device_columns: `columns` as for Writer; values[name] /
        nulls[name] are DEVICE addresses (e.g. torch tensor .data_ptr()) of num_rows
        u64 value words / NULL-flag bytes (nulls only for optional columns);
        heaps[name]: byte heap of a STRING_PLAIN column, whose value words are
        (length << 40) | offset into it; levels[name] = (rlevels address | None,
        dlevels address, num_slots) for a repeated / nested column, whose values are
        given per slot; page_order: capi.PAGE_ORDER_COLUMNS / PAGE_ORDER_ROWS"""
        names = [c["name"].encode() for c in columns]
        specs = (K.ColumnSpec * len(columns))()
        data = (K.DeviceColumn * len(columns))()
        for i, c in enumerate(columns):
            specs[i] = K.ColumnSpec(names[i], c["logical_type"], c["storage_type"],
                                    c.get("column_id", i + 1), c.get("rlevel_max", 0),
                                    c.get("dlevel_max", 0), c.get("bitpack_max_value", 0))
            rl, dl, ns = (levels or {}).get(c["name"], (None, None, 0))
            data[i] = K.DeviceColumn(values[c["name"]], (nulls or {}).get(c["name"]),
                                     (heaps or {}).get(c["name"]), rl, dl, ns)
        t = C.c_void_p()
        _check(lib().evql_table_from_device_columns_ordered(self.h, specs, len(columns), data,
                                                            num_rows, page_order, C.byref(t)))
        return Table(self, t)

    def close(self):
        if self.h:
            lib().evql_ctx_destroy(self.h)
            self.h = None


class Table:
    def __init__(self, ctx, h):
        self.ctx = ctx
        self.h = h

    @property
    def num_rows(self):
        return lib().evql_table_num_rows(self.h)

    def columns(self):
        out = []
        for i in range(lib().evql_table_num_columns(self.h)):
            ci = K.ColumnInfo()
            _check(lib().evql_table_column_info(self.h, i, C.byref(ci)))
            out.append(dict(name=ci.name.decode(), logical_type=ci.logical_type,
                            storage_type=ci.storage_type, column_id=ci.column_id,
                            rlevel_max=ci.rlevel_max, dlevel_max=ci.dlevel_max,
                            n_data_pages=ci.n_data_pages, payload_bytes=ci.payload_bytes))
        return out

    def schema(self):
        """column name -> evql_stype as FastCSTableScan would type it"""
        from .plan import stype_of_column
        return {c["name"]: stype_of_column(c["logical_type"]) for c in self.columns()
                if c["logical_type"] != K.COL_SUBRECORD}

    def download_image(self):
        n = lib().evql_table_image_size(self.h)
        buf = C.create_string_buffer(n)
        _check(lib().evql_table_download_image(self.h, buf, n))
        return buf.raw

    def query(self, plan):
        q = C.c_void_p()
        _check(lib().evql_query_create(self.ctx.h, self.h, C.byref(plan.desc), C.byref(q)))
        return Query(self, plan, q)

    def close(self):
        if self.h:
            lib().evql_table_close(self.h)
            self.h = None


class Result:
    def __init__(self, columns, types, raw):
        self.columns = columns
        self.types = types
        self.raw = raw
        self.nrows = len(columns[0]) if columns else 0

    def rows(self):
        return list(zip(*self.columns)) if self.columns else []


class Query:
    """mirrors csql::TableExpression: execute() once, next_batch() until 0 rows"""

    def __init__(self, table, plan, h):
        self.table = table
        self.plan = plan  # keeps the ctypes buffers alive
        self.h = h

    def set_order(self, order):
        """fuse ORDER BY .. LIMIT (eventql_amd.plan.Order) above the GROUP BY"""
        self.order = order  # keeps the ctypes buffers alive
        _check(lib().evql_query_set_order(self.h, order.specs, order.n, order.limit,
                                          order.offset))

    def execute(self):
        _check(lib().evql_query_execute(self.h, None, None))

    def launch(self):
        _check(lib().evql_query_launch(self.h))

    def finish(self):
        _check(lib().evql_query_finish(self.h))

    def column_count(self):
        return lib().evql_query_column_count(self.h)

    def column_type(self, i):
        return lib().evql_query_column_type(self.h, i)

    def next_batch(self, max_rows=1024):
        nc = self.column_count()
        bufs = (K.ColumnBuf * max(1, nc))()
        n = C.c_size_t()
        _check(lib().evql_query_next_batch(self.h, max_rows, bufs, C.byref(n)))
        raw = [C.string_at(bufs[i].data, bufs[i].size) if bufs[i].size else b""
               for i in range(nc)]
        return n.value, raw

    def fetch_all(self, batch=1024):
        nc = self.column_count()
        types = [self.column_type(i) for i in range(nc)]
        chunks = [[] for _ in range(nc)]
        while True:
            n, raw = self.next_batch(batch)
            if n == 0:
                break
            for c, r in zip(chunks, raw):
                c.append(r)
        raws = [b"".join(c) for c in chunks]
        cols = [unpack_svector(t, r) for t, r in zip(types, raws)]
        return Result(cols, types, raws)

    def run(self):
        self.execute()
        return self.fetch_all()

    def stats(self):
        s = K.QueryStats()
        _check(lib().evql_query_stats(self.h, C.byref(s)))
        return {f[0]: getattr(s, f[0]) for f in K.QueryStats._fields_}

    def kernel_source(self):
        return lib().evql_query_kernel_source(self.h).decode()

    def record_words(self):
        return lib().evql_query_record_words(self.h)

    def export_groups(self, device_ptr, max_groups):
        n = C.c_uint64()
        _check(lib().evql_query_export_groups(self.h, device_ptr, max_groups, C.byref(n)))
        return n.value

    def import_groups(self, device_ptr, n):
        _check(lib().evql_query_import_groups(self.h, device_ptr, n))

    def export_pairs(self, which, device_ptr, max_pairs):
        """count_distinct pairs of the which-th such aggregate (3 words each); device_ptr
        None: the count only"""
        n = C.c_uint64()
        _check(lib().evql_query_export_pairs(self.h, which, device_ptr, max_pairs, C.byref(n)))
        return n.value

    def import_pairs(self, which, device_ptr, n):
        _check(lib().evql_query_import_pairs(self.h, which, device_ptr, n))

    def exchange(self, x, mode=K.EXCHANGE_GATHER_ALL):
        """evql_query_exchange (collective): afterwards next_batch yields the merged
        groups (all of them, or with EXCHANGE_BY_OWNER the key range this rank owns)"""
        _check(lib().evql_query_exchange(self.h, x.h, mode))

    def reset(self):
        """empty group table, no scan: merge target for import_groups"""
        _check(lib().evql_query_reset(self.h))

    def close(self):
        if self.h:
            lib().evql_query_destroy(self.h)
            self.h = None


class LsmChain:
    """row filters of a newest-first chain of LSM tables, built on the device
    (PartitionCursor::openNextTable, partition_cursor.cc:160-195)"""

    def __init__(self, ctx):
        self.ctx = ctx
        self.tables = []
        self.h = C.c_void_p()
        _check(lib().evql_lsm_chain_create(ctx.h, C.byref(self.h)))

    def add(self, table, has_skiplist=False, arena_skiplist=None, has_updates=True):
        """tables in scan order (newest first); has_skiplist / has_updates are the file's
        LSMTableRef flags, arena_skiplist (one byte per row) marks an arena"""
        sk = None if arena_skiplist is None else np.ascontiguousarray(arena_skiplist, np.uint8)
        flags = (K.LSM_HAS_SKIPLIST if has_skiplist else 0) | (K.LSM_HAS_UPDATES if has_updates else 0)
        _check(lib().evql_lsm_chain_add(self.h, table.h, flags, _ptr(sk, _u8p),
                                        0 if sk is None else len(sk)))
        self.tables.append(table)

    def build(self):
        _check(lib().evql_lsm_chain_build(self.h))

    def filter(self, idx):
        """(bool array per row -- None when the table needs no filter --, rows kept) of
        the idx-th table"""
        bits = _u8p()
        n = C.c_uint64()
        kept = C.c_uint64()
        _check(lib().evql_lsm_chain_filter(self.h, idx, C.byref(bits), C.byref(n), C.byref(kept)))
        if not bits:
            return None, kept.value
        raw = np.frombuffer(C.string_at(bits, (n.value + 63) // 64 * 8), np.uint8)
        return np.unpackbits(raw, bitorder="little")[:n.value].astype(bool), kept.value

    def query(self, plan):
        """the operator over the whole chain (GroupByExpression over PartitionCursor)"""
        q = C.c_void_p()
        _check(lib().evql_query_create_chain(self.ctx.h, self.h, C.byref(plan.desc), C.byref(q)))
        return Query(self, plan, q)

    def close(self):
        if self.h:
            lib().evql_lsm_chain_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Merge:
    """mirrors csql::GroupByMergeExpression (groupby.cc:493-672) for partial
    aggregates that arrive as bytes: add_frame()/add_rows(), then fetch_all()"""

    def __init__(self, plan):
        self.plan = plan
        self.h = C.c_void_p()
        _check(lib().evql_merge_create(C.byref(plan.desc), C.byref(self.h)))

    def add_frame(self, payload):
        buf = bytes(payload)
        _check(lib().evql_merge_add_frame(self.h, buf, len(buf)))

    def add_rows(self, keys_raw, data_raw, nrows):
        _check(lib().evql_merge_add_rows(self.h, keys_raw, len(keys_raw), data_raw,
                                         len(data_raw), nrows))

    @property
    def num_groups(self):
        return lib().evql_merge_num_groups(self.h)

    def fetch_all(self, batch=1024):
        L = lib()
        nc = L.evql_merge_column_count(self.h)
        types = [L.evql_merge_column_type(self.h, i) for i in range(nc)]
        raws = [b""] * nc
        bufs = (K.ColumnBuf * max(1, nc))()
        n = C.c_size_t()
        while True:
            _check(L.evql_merge_next_batch(self.h, batch, bufs, C.byref(n)))
            if n.value == 0:
                break
            raws = [raws[i] + (C.string_at(bufs[i].data, bufs[i].size) if bufs[i].size else b"")
                    for i in range(nc)]
        return Result([unpack_svector(t, r) for t, r in zip(types, raws)], types, raws)

    def close(self):
        if self.h:
            lib().evql_merge_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def compile_only(plan, columns, cache_dir=None):
    """compile the fused kernel of `plan` for gfx950 without a device.
    columns: list of dict(name, logical_type, storage_type, dlevel_max=0, bits=0)"""
    infos = (K.ColumnInfo * len(columns))()
    for i, c in enumerate(columns):
        infos[i].name = c["name"].encode()
        infos[i].logical_type = c["logical_type"]
        infos[i].storage_type = c["storage_type"]
        infos[i].column_id = i + 1
        infos[i].dlevel_max = c.get("dlevel_max", 0)
        infos[i].rlevel_max = c.get("rlevel_max", 0)
        infos[i].payload_bytes = c.get("bits", 0)
    size = C.c_size_t()
    cd = (cache_dir or KERNEL_CACHE_DIR).encode()
    _check(lib().evql_compile_only(C.byref(plan.desc), infos, len(columns), cd, C.byref(size)))
    return size.value


class Hub:
    """evql_hub_*: the in-process rendezvous of nranks contexts driven by one thread each"""

    def __init__(self, nranks):
        self.h = C.c_void_p()
        _check(lib().evql_hub_create(nranks, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().evql_hub_destroy(self.h)
            self.h = None


class Exchange:
    """evql_exchange_*: moves group records between the ranks of a distributed GROUP BY.
    Exchange.hub(ctx, hub, rank) | Exchange.rccl(ctx, nranks, rank, id128) |
    Exchange.custom(ctx, nranks, rank, all_gather, all_to_all, name)"""

    def __init__(self, h, keep=None):
        self.h = h
        self._keep = keep

    @classmethod
    def hub(cls, ctx, hub, rank):
        h = C.c_void_p()
        _check(lib().evql_exchange_create_hub(ctx.h, hub.h, rank, C.byref(h)))
        return cls(h)

    @staticmethod
    def rccl_unique_id():
        buf = C.create_string_buffer(128)
        _check(lib().evql_rccl_unique_id(buf))
        return buf.raw

    @classmethod
    def rccl(cls, ctx, nranks, rank, id128):
        h = C.c_void_p()
        _check(lib().evql_exchange_create_rccl(ctx.h, nranks, rank, bytes(id128), C.byref(h)))
        return cls(h)

    @classmethod
    def custom(cls, ctx, nranks, rank, all_gather, all_to_all, name="custom"):
        """all_gather(send: list[int]) -> list[int] (nranks * len(send));
        all_to_all(d_send, send_counts, d_recv, recv_counts, stream) with raw device
        addresses and word counts"""
        def ag(user, send, n, recv):
            try:
                out = all_gather([send[i] for i in range(n)])
                for i, v in enumerate(out):
                    recv[i] = v
                return 0
            except Exception:  # noqa: BLE001 -- must not unwind through C
                import traceback
                traceback.print_exc()
                return K.EVQL_ERUNTIME

        def a2a(user, d_send, sc, d_recv, rc, stream):
            try:
                all_to_all(d_send, [sc[i] for i in range(nranks)], d_recv,
                           [rc[i] for i in range(nranks)], stream)
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return K.EVQL_ERUNTIME

        cb1, cb2 = K.ALL_GATHER_FN(ag), K.ALL_TO_ALL_FN(a2a)
        tr = K.Transport(None, cb1, cb2, name.encode())
        h = C.c_void_p()
        _check(lib().evql_exchange_create(ctx.h, nranks, rank, C.byref(tr), C.byref(h)))
        return cls(h, keep=(cb1, cb2, tr))

    def backend(self):
        return lib().evql_exchange_backend(self.h).decode()

    def stats(self):
        s = K.ExchangeStats()
        _check(lib().evql_exchange_last_stats(self.h, C.byref(s)))
        return {f[0]: getattr(s, f[0]) for f in K.ExchangeStats._fields_}

    def close(self):
        if self.h:
            lib().evql_exchange_destroy(self.h)
            self.h = None
