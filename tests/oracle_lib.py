"""ctypes bindings of the TEST-ONLY libraries:

  oracle/_build/liboracle.so    CPU restatement of the reference algorithm
  oracle/_ref/libcstable_ref.so the reference's own cstable library (compiled
                                from /root/reference in place; optional)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from eventql_amd import capi as K
from eventql_amd.plan import unpack_svector

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# EVQL_ORACLE_SO: an instrumented build (oracle/Makefile `asan`), run with
# LD_PRELOAD=$(gcc -print-file-name=libasan.so)
ORACLE_SO = os.environ.get("EVQL_ORACLE_SO") or os.path.join(ORACLE_DIR, "_build", "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libcstable_ref.so")

_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)
_f64p = C.POINTER(C.c_double)


def _np_ptr(a, ty):
    return a.ctypes.data_as(ty) if a is not None else None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.orc_last_error.restype = C.c_char_p
        L.orc_query_error.restype = C.c_char_p
        L.orc_table_open.restype = C.c_void_p
        L.orc_table_open.argtypes = [C.c_char_p]
        L.orc_table_open_image.restype = C.c_void_p
        L.orc_table_open_image.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_table_close.argtypes = [C.c_void_p]
        L.orc_table_version.argtypes = [C.c_void_p]
        L.orc_table_num_rows.restype = C.c_uint64
        L.orc_table_num_rows.argtypes = [C.c_void_p]
        L.orc_table_num_columns.argtypes = [C.c_void_p]
        L.orc_table_column_info.argtypes = [
            C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
            _u64p, _u64p, _u64p]
        L.orc_table_column_num_values.restype = C.c_uint64
        L.orc_table_column_num_values.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_column_open.restype = C.c_void_p
        L.orc_column_open.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_column_close.argtypes = [C.c_void_p]
        L.orc_column_read_uint.argtypes = [C.c_void_p, C.c_uint64, _u64p, _u64p, _u8p, _u64p]
        L.orc_column_read_float.argtypes = [C.c_void_p, C.c_uint64, _u64p, _u64p, _u8p, _f64p]
        L.orc_column_read_string.argtypes = [
            C.c_void_p, C.c_uint64, _u64p, _u64p, _u8p, _u64p, C.c_char_p, C.c_uint64]
        L.orc_query_run.restype = C.c_void_p
        L.orc_query_run.argtypes = [C.c_void_p, C.POINTER(K.PlanDesc)]
        L.orc_result_free.argtypes = [C.c_void_p]
        L.orc_result_num_columns.argtypes = [C.c_void_p]
        L.orc_result_column_type.argtypes = [C.c_void_p, C.c_int]
        L.orc_result_num_rows.restype = C.c_uint64
        L.orc_result_num_rows.argtypes = [C.c_void_p]
        L.orc_result_column_data.restype = C.POINTER(C.c_uint8)
        L.orc_result_column_data.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
        L.orc_result_group_keys.restype = C.POINTER(C.c_uint8)
        L.orc_result_group_keys.argtypes = [C.c_void_p]
        L.orc_result_rows_scanned.restype = C.c_uint64
        L.orc_result_rows_scanned.argtypes = [C.c_void_p]
        L.orc_result_rows_passed.restype = C.c_uint64
        L.orc_result_rows_passed.argtypes = [C.c_void_p]
        L.orc_sha1.argtypes = [C.c_void_p, C.c_size_t, _u8p]
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


_ref = None


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.ref_last_error.restype = C.c_char_p
        L.ref_writer_create.restype = C.c_void_p
        L.ref_writer_create.argtypes = [
            C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int),
            C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        for fn, vt in (("ref_writer_put_uint", _u64p), ("ref_writer_put_float", _f64p)):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, _u64p, _u64p, _u8p, vt]
        L.ref_writer_put_string.argtypes = [
            C.c_void_p, C.c_char_p, C.c_uint64, _u64p, _u64p, _u8p, _u64p, C.c_char_p]
        L.ref_writer_commit.argtypes = [C.c_void_p, C.c_uint64]
        L.ref_writer_free.argtypes = [C.c_void_p]
        L.ref_reader_open.restype = C.c_void_p
        L.ref_reader_open.argtypes = [C.c_char_p]
        L.ref_reader_free.argtypes = [C.c_void_p]
        L.ref_reader_num_records.restype = C.c_uint64
        L.ref_reader_num_records.argtypes = [C.c_void_p]
        L.ref_reader_num_columns.argtypes = [C.c_void_p]
        L.ref_reader_column_info.argtypes = [
            C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
            _u64p, _u64p, _u64p]
        L.ref_column_open.restype = C.c_void_p
        L.ref_column_open.argtypes = [C.c_void_p, C.c_char_p]
        L.ref_column_close.argtypes = [C.c_void_p]
        L.ref_column_read_uint.argtypes = [C.c_void_p, C.c_uint64, _u64p, _u64p, _u8p, _u64p]
        L.ref_column_read_float.argtypes = [C.c_void_p, C.c_uint64, _u64p, _u64p, _u8p, _f64p]
        L.ref_column_read_string.argtypes = [
            C.c_void_p, C.c_uint64, _u64p, _u64p, _u8p, _u64p, C.c_char_p, C.c_uint64]
        L.ref_sha1.argtypes = [C.c_void_p, C.c_uint64, _u8p]
        _ref = L
    return _ref


# ---------------------------------------------------------------------------
# generic reader facade: works for the oracle ("orc") and the reference ("ref")
# ---------------------------------------------------------------------------
class TableReader:
    def __init__(self, path, which="orc"):
        self.which = which
        if which == "orc":
            self.L = oracle()
            self.h = self.L.orc_table_open(path.encode())
            if not self.h:
                raise IOError(self.L.orc_last_error().decode())
            self.num_rows = self.L.orc_table_num_rows(self.h)
            self.ncols = self.L.orc_table_num_columns(self.h)
            self._info = self.L.orc_table_column_info
            self._open = self.L.orc_column_open
            self._close = self.L.orc_column_close
            self._pfx = "orc_column_"
        else:
            self.L = ref()
            self.h = self.L.ref_reader_open(path.encode())
            if not self.h:
                raise IOError(self.L.ref_last_error().decode())
            self.num_rows = self.L.ref_reader_num_records(self.h)
            self.ncols = self.L.ref_reader_num_columns(self.h)
            self._info = self.L.ref_reader_column_info
            self._open = self.L.ref_column_open
            self._close = self.L.ref_column_close
            self._pfx = "ref_column_"

    def columns(self):
        out = []
        for i in range(self.ncols):
            name = C.create_string_buffer(256)
            lt, st = C.c_int(), C.c_int()
            cid, rm, dm = C.c_uint64(), C.c_uint64(), C.c_uint64()
            rc = self._info(self.h, i, name, C.byref(lt), C.byref(st), C.byref(cid),
                            C.byref(rm), C.byref(dm))
            assert rc == 0
            out.append(dict(name=name.value.decode(), logical_type=lt.value,
                            storage_type=st.value, column_id=cid.value,
                            rlevel_max=rm.value, dlevel_max=dm.value))
        return out

    def read(self, name, n, kind):
        """returns (rlvl, dlvl, present, values); kind in uint|float|string"""
        c = self._open(self.h, name.encode())
        if not c:
            raise KeyError(name)
        rl = np.zeros(n, np.uint64)
        dl = np.zeros(n, np.uint64)
        pr = np.zeros(n, np.uint8)
        try:
            if kind == "uint":
                v = np.zeros(n, np.uint64)
                rc = getattr(self.L, self._pfx + "read_uint")(
                    c, n, _np_ptr(rl, _u64p), _np_ptr(dl, _u64p), _np_ptr(pr, _u8p),
                    _np_ptr(v, _u64p))
                assert rc == 0
                return rl, dl, pr, v
            if kind == "float":
                v = np.zeros(n, np.float64)
                rc = getattr(self.L, self._pfx + "read_float")(
                    c, n, _np_ptr(rl, _u64p), _np_ptr(dl, _u64p), _np_ptr(pr, _u8p),
                    _np_ptr(v, _f64p))
                assert rc == 0
                return rl, dl, pr, v
            cap = 1 << 20
            while True:
                off = np.zeros(n + 1, np.uint64)
                buf = C.create_string_buffer(cap)
                rc = getattr(self.L, self._pfx + "read_string")(
                    c, n, _np_ptr(rl, _u64p), _np_ptr(dl, _u64p), _np_ptr(pr, _u8p),
                    _np_ptr(off, _u64p), buf, cap)
                if rc == -2:
                    self._close(c)
                    c = self._open(self.h, name.encode())
                    cap *= 4
                    continue
                assert rc == 0
                raw = buf.raw
                vals = [raw[int(off[i]):int(off[i + 1])] for i in range(n)]
                return rl, dl, pr, vals
        finally:
            self._close(c)

    def close(self):
        if self.h:
            if self.which == "orc":
                self.L.orc_table_close(self.h)
            else:
                self.L.ref_reader_free(self.h)
            self.h = None


def ref_write_table(path, schema_nodes, columns, num_rows):
    """write a table with the REFERENCE writer.

    schema_nodes: list of dict(name, type, encoding, repeated, optional, parent)
    columns: list of (flat_name, kind, values, rlvl|None, dlvl|None, present|None)
             for kind == 'string' values = list of bytes
    """
    L = ref()
    n = len(schema_nodes)
    names = (C.c_char_p * n)(*[s["name"].encode() for s in schema_nodes])
    arr = lambda k: (C.c_int * n)(*[int(s[k]) for s in schema_nodes])
    w = L.ref_writer_create(path.encode(), n, names, arr("type"), arr("encoding"),
                            arr("repeated"), arr("optional"), arr("parent"))
    if not w:
        raise IOError(L.ref_last_error().decode())
    try:
        for (name, kind, values, rl, dl, pr) in columns:
            cnt = len(values)
            rlp = _np_ptr(np.ascontiguousarray(rl, np.uint64), _u64p) if rl is not None else None
            dlp = _np_ptr(np.ascontiguousarray(dl, np.uint64), _u64p) if dl is not None else None
            prp = _np_ptr(np.ascontiguousarray(pr, np.uint8), _u8p) if pr is not None else None
            if kind == "uint":
                v = np.ascontiguousarray(values, np.uint64)
                rc = L.ref_writer_put_uint(w, name.encode(), cnt, rlp, dlp, prp, _np_ptr(v, _u64p))
            elif kind == "float":
                v = np.ascontiguousarray(values, np.float64)
                rc = L.ref_writer_put_float(w, name.encode(), cnt, rlp, dlp, prp, _np_ptr(v, _f64p))
            else:
                off = np.zeros(cnt + 1, np.uint64)
                off[1:] = np.cumsum([len(s) for s in values])
                blob = b"".join(values)
                rc = L.ref_writer_put_string(w, name.encode(), cnt, rlp, dlp, prp,
                                             _np_ptr(off, _u64p), blob)
            if rc != 0:
                raise IOError(L.ref_last_error().decode())
        if L.ref_writer_commit(w, num_rows) != 0:
            raise IOError(L.ref_last_error().decode())
    finally:
        L.ref_writer_free(w)


def ref_write_table_by_records(path, schema_nodes, columns, num_records):
    """REFERENCE writer, record by record: every column per record in the order given
    (the call order of a row-wise writer, ref_shim.cc ref_writer_put_records).
    columns: list of (flat_name, 'uint'|'float', value words (u64 / f64 bits),
             rlvl|None, dlvl|None, present|None), one entry per slot"""
    L = ref()
    L.ref_writer_put_records.argtypes = [
        C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), _u64p,
        C.POINTER(_u64p), C.POINTER(_u64p), C.POINTER(_u8p), C.POINTER(_u64p), C.c_uint64]
    n = len(schema_nodes)
    names = (C.c_char_p * n)(*[s["name"].encode() for s in schema_nodes])
    arr = lambda k: (C.c_int * n)(*[int(s[k]) for s in schema_nodes])
    w = L.ref_writer_create(path.encode(), n, names, arr("type"), arr("encoding"),
                            arr("repeated"), arr("optional"), arr("parent"))
    if not w:
        raise IOError(L.ref_last_error().decode())
    try:
        nc = len(columns)
        keep = []

        def ptr(a, dt, pt):
            if a is None:
                return pt()
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            return _np_ptr(a, pt)
        cnames = (C.c_char_p * nc)(*[c[0].encode() for c in columns])
        kinds = (C.c_int * nc)(*[1 if c[1] == "float" else 0 for c in columns])
        nslots = np.array([len(c[2]) for c in columns], np.uint64)
        vals = (_u64p * nc)(*[ptr(np.asarray(c[2]).view(np.uint64), np.uint64, _u64p) for c in columns])
        rls = (_u64p * nc)(*[ptr(c[3], np.uint64, _u64p) for c in columns])
        dls = (_u64p * nc)(*[ptr(c[4], np.uint64, _u64p) for c in columns])
        prs = (_u8p * nc)(*[ptr(c[5], np.uint8, _u8p) for c in columns])
        rc = L.ref_writer_put_records(w, nc, cnames, kinds, _np_ptr(nslots, _u64p), rls, dls, prs,
                                      vals, num_records)
        if rc != 0:
            raise IOError(L.ref_last_error().decode())
        if L.ref_writer_commit(w, num_records) != 0:
            raise IOError(L.ref_last_error().decode())
    finally:
        L.ref_writer_free(w)


# ---------------------------------------------------------------------------
# oracle query runner
# ---------------------------------------------------------------------------
class OracleResult:
    def __init__(self, columns, types, nrows, keys, rows_scanned, rows_passed, raw):
        self.columns = columns  # list of python value lists
        self.types = types
        self.nrows = nrows
        self.keys = keys
        self.rows_scanned = rows_scanned
        self.rows_passed = rows_passed
        self.raw = raw          # packed SVector bytes per column

    def rows(self):
        return list(zip(*self.columns)) if self.columns else []


def oracle_run(path_or_image, plan, order=None):
    """order: eventql_amd.plan.Order -> OrderByExpression + LimitExpression applied
    to the operator's output (orc_result_order_limit)"""
    L = oracle()
    if isinstance(path_or_image, (bytes, bytearray, memoryview)):
        buf = bytes(path_or_image)
        t = L.orc_table_open_image(buf, len(buf))
    else:
        t = L.orc_table_open(path_or_image.encode())
    if not t:
        raise IOError(L.orc_last_error().decode())
    try:
        r = L.orc_query_run(t, C.byref(plan.desc))
        if not r:
            raise RuntimeError(L.orc_query_error().decode())
        try:
            if order is not None:
                L.orc_result_order_limit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32,
                                                     C.c_int64, C.c_uint64]
                if L.orc_result_order_limit(r, C.cast(order.specs, C.c_void_p), order.n,
                                            order.limit, order.offset):
                    raise RuntimeError(L.orc_query_error().decode())
            return _collect(L, r)
        finally:
            L.orc_result_free(r)
    finally:
        L.orc_table_close(t)


def oracle_time_parallel(path, plans):
    """wall-clock seconds of len(plans) concurrent orc_query_run calls over the same
    file, one thread each (ctypes releases the GIL; tables are opened and results
    freed outside the timed region) -- bench.py's all-cores CPU figure"""
    import threading
    import time
    L = oracle()
    tables = [L.orc_table_open(path.encode()) for _ in plans]
    if not all(tables):
        raise IOError(L.orc_last_error().decode())
    results = [None] * len(plans)

    def work(i):
        results[i] = L.orc_query_run(tables[i], C.byref(plans[i].desc))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(plans))]
    t0 = time.time()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt = time.time() - t0
    ok = all(results)
    for r in results:
        if r:
            L.orc_result_free(r)
    for t in tables:
        L.orc_table_close(t)
    if not ok:
        raise RuntimeError("oracle run failed")
    return dt


def _collect(L, r):
    nc = L.orc_result_num_columns(r)
    nrows = L.orc_result_num_rows(r)
    cols, types, raws = [], [], []
    for i in range(nc):
        sz = C.c_size_t()
        p = L.orc_result_column_data(r, i, C.byref(sz))
        raw = C.string_at(p, sz.value) if sz.value else b""
        ty = L.orc_result_column_type(r, i)
        types.append(ty)
        raws.append(raw)
        cols.append(unpack_svector(ty, raw))
    keys = None
    kp = L.orc_result_group_keys(r)
    if kp:
        keys = C.string_at(kp, 20 * nrows)
    return OracleResult(cols, types, nrows, keys, L.orc_result_rows_scanned(r),
                        L.orc_result_rows_passed(r), raws)


def varuint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def partial_frame(keys, datas, flags=0):
    """QUERY_PARTIALAGGR_RESULT payload (frames/query_partialaggr_result.cc:53-57):
    varuint flags, varuint num_rows, then (20-B key, data) per row"""
    return varuint(flags) + varuint(len(keys)) + b"".join(k + d for k, d in zip(keys, datas))


def oracle_partial_frame(path_or_image, plan):
    """runs `plan` (EVQL_MODE_PARTIAL) through the oracle and frames its rows"""
    r = oracle_run(path_or_image, plan)
    keys = [r.keys[20 * i:20 * i + 20] for i in range(r.nrows)]
    return partial_frame(keys, r.columns[0])


def oracle_lsm_filters(images, has_skip_column, arena_skips=None):
    """PartitionCursor::openNextTable restatement over a newest-first chain of table
    images: list of bool arrays (row scanned?)"""
    import numpy as np
    L = oracle()
    L.orc_lsm_create.restype = C.c_void_p
    L.orc_lsm_free.argtypes = [C.c_void_p]
    L.orc_lsm_next_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_char_p]
    m = L.orc_lsm_create()
    out = []
    try:
        for i, img in enumerate(images):
            buf = bytes(img)
            t = L.orc_table_open_image(buf, len(buf))
            if not t:
                raise IOError(L.orc_last_error().decode())
            try:
                n = L.orc_table_num_rows(t)
                res = C.create_string_buffer(max(1, n))
                sk = None
                if arena_skips is not None and arena_skips[i] is not None:
                    sk = np.ascontiguousarray(arena_skips[i], np.uint8).tobytes()
                rc = L.orc_lsm_next_table(m, t, int(has_skip_column[i]), sk, res)
                if rc == -2:
                    raise RuntimeError("invalid SHA1Hash")
                if rc:
                    raise IOError("oracle lsm read error")
                out.append(np.frombuffer(res.raw[:n], np.uint8).astype(bool))
            finally:
                L.orc_table_close(t)
    finally:
        L.orc_lsm_free(m)
    return out


def oracle_partition_filters(files_oldest_first):
    """PartitionCursor::openNextTable over the LSM files of a partition
    (tests/lsm_tables.partition): bool array per file in SCAN order (newest first), None
    where the cursor does not call setFilter (orc_lsm_next_file)"""
    import numpy as np
    L = oracle()
    L.orc_lsm_create.restype = C.c_void_p
    L.orc_lsm_free.argtypes = [C.c_void_p]
    L.orc_lsm_next_file.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p,
                                    C.c_char_p, C.POINTER(C.c_int)]
    m = L.orc_lsm_create()
    out = []
    try:
        n = len(files_oldest_first)
        for k, (_, img, skl, upd, _) in enumerate(reversed(files_oldest_first)):
            buf = bytes(img)
            t = L.orc_table_open_image(buf, len(buf))
            if not t:
                raise IOError(L.orc_last_error().decode())
            try:
                rows = L.orc_table_num_rows(t)
                res = C.create_string_buffer(max(1, rows))
                needs = C.c_int(0)
                rc = L.orc_lsm_next_file(m, t, int(skl), int(upd), int(n - 1 - k == 0), None, res,
                                         C.byref(needs))
                if rc == -2:
                    raise RuntimeError("invalid SHA1Hash")
                if rc:
                    raise IOError("oracle lsm read error")
                out.append(np.frombuffer(res.raw[:rows], np.uint8).astype(bool)
                           if needs.value else None)
            finally:
                L.orc_table_close(t)
    finally:
        L.orc_lsm_free(m)
    return out


def oracle_run_chain(images_scan_order, filters, plan):
    """GroupByExpression / bare scan over PartitionCursor (orc_query_run_chain): the
    tables in scan order, filters[i] a bool array or None"""
    import numpy as np
    L = oracle()
    n = len(images_scan_order)
    bufs = [bytes(i) for i in images_scan_order]
    tabs = [L.orc_table_open_image(b, len(b)) for b in bufs]
    if not all(tabs):
        raise IOError(L.orc_last_error().decode())
    packed = [None if f is None else np.packbits(np.asarray(f, bool), bitorder="little").tobytes() + b"\0"
              for f in filters]
    L.orc_query_run_chain.restype = C.c_void_p
    L.orc_query_run_chain.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_char_p),
                                      C.POINTER(C.c_uint64), C.c_void_p]
    tarr = (C.c_void_p * n)(*tabs)
    farr = (C.c_char_p * n)(*packed)
    larr = (C.c_uint64 * n)(*[0 if f is None else len(f) for f in filters])
    try:
        r = L.orc_query_run_chain(tarr, n, farr, larr, C.byref(plan.desc))
        if not r:
            raise RuntimeError(L.orc_query_error().decode())
        try:
            return _collect(L, r)
        finally:
            L.orc_result_free(r)
    finally:
        for t in tabs:
            L.orc_table_close(t)


def oracle_merge(plan, frames):
    """GroupByMergeExpression restatement over frame payloads"""
    L = oracle()
    L.orc_merge_frames.restype = C.c_void_p
    L.orc_merge_frames.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t),
                                   C.c_int]
    arr = (C.c_char_p * len(frames))(*frames)
    lens = (C.c_size_t * len(frames))(*[len(f) for f in frames])
    r = L.orc_merge_frames(C.byref(plan.desc), arr, lens, len(frames))
    if not r:
        raise RuntimeError(L.orc_query_error().decode())
    try:
        return _collect(L, r)
    finally:
        L.orc_result_free(r)


def sha1(data, which="orc"):
    out = (C.c_uint8 * 20)()
    if which == "orc":
        oracle().orc_sha1(data, len(data), out)
    else:
        ref().ref_sha1(data, len(data), out)
    return bytes(out)
