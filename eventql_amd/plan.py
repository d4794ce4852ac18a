"""Expression trees -> csql bytecode -> evql_plan_desc_t.

This module plays the role that the reference's query-tree + compiler play for
the hot path: it produces exactly the `vm::Program` instruction stream the
reference's `Compiler::compile` (sql/runtime/compiler.cc:50-248) would produce
for the same expression, so that tests and benchmarks can drive the C ABI (and
the oracle) without the SQL parser:

  * arguments are compiled left to right, the call comes last (post-order)
  * IF(c,t,f) = c; CJUMP ->T; f; JUMP ->end; T: t           (compiler.cc:174-209)
  * an aggregate expression compiles to
      [method_call:  ... X_CALL_INSTANCE get ...; RETURN]
      [method_accumulate: args...; X_CALL_INSTANCE accumulate; RETURN]
    (compiler.cc:57-100); only the FIRST aggregate sub-expression gets an
    instance (QueryTreeUtil::findAggregateExpression, qtree/QueryTreeUtil.cc:209-224)
  * literals: non-negative int -> UINT64, negative int -> INT64, float ->
    FLOAT64 (runtime/queryplanbuilder.cc:1519-1529)
  * implicit conversions UINT64->INT64 and X->NIL insert to_int64 / to_nil
    calls (qtree/CallExpressionNode.cc:58-85, defaults.cc:39-46)
  * the GROUP BY level reads the scan's output columns, which are bare column
    references appended on first use (qtree/SequentialScanNode.cc:211-238)
"""
import ctypes as C
import struct

from . import capi as K

_TS = {K.T_UINT64: K.TS_UINT64, K.T_INT64: K.TS_INT64, K.T_FLOAT64: K.TS_FLOAT64,
       K.T_BOOL: K.TS_BOOL, K.T_STRING: K.TS_STRING,
       K.T_TIMESTAMP64: K.TS_TIMESTAMP64, K.T_NIL: K.TS_NIL}

_CMP_FAMS = {"cmp": K.FAM_CMP, "eq": K.FAM_EQ, "neq": K.FAM_NEQ, "lt": K.FAM_LT,
             "lte": K.FAM_LTE, "gt": K.FAM_GT, "gte": K.FAM_GTE}
_ARITH_FAMS = {"add": K.FAM_ADD, "sub": K.FAM_SUB, "mul": K.FAM_MUL,
               "div": K.FAM_DIV, "mod": K.FAM_MOD, "pow": K.FAM_POW}
# string -> string functions of expressions/string.cc (lowercase / uppercase are aliases)
_STR1_FAMS = {"lcase": K.FAM_LCASE, "lowercase": K.FAM_LCASE, "ucase": K.FAM_UCASE,
              "uppercase": K.FAM_UCASE, "ltrim": K.FAM_LTRIM, "rtrim": K.FAM_RTRIM}


class Expr:
    rtype = K.T_NIL

    # sugar ---------------------------------------------------------------
    def _bin(self, name, other):
        return Call(name, self, _wrap(other))

    def __gt__(self, o): return self._bin("gt", o)
    def __ge__(self, o): return self._bin("gte", o)
    def __lt__(self, o): return self._bin("lt", o)
    def __le__(self, o): return self._bin("lte", o)
    def eq(self, o): return self._bin("eq", o)
    def neq(self, o): return self._bin("neq", o)
    def __add__(self, o): return self._bin("add", o)
    def __sub__(self, o): return self._bin("sub", o)
    def __mul__(self, o): return self._bin("mul", o)
    def __truediv__(self, o): return self._bin("div", o)
    def __mod__(self, o): return self._bin("mod", o)
    def __and__(self, o): return Call("logical_and", self, _wrap(o))
    def __or__(self, o): return Call("logical_or", self, _wrap(o))
    def __invert__(self): return Call("neg", self)


def _wrap(v):
    return v if isinstance(v, Expr) else Lit(v)


class Col(Expr):
    def __init__(self, name, rtype=None):
        self.name = name
        self.rtype = rtype  # resolved against the schema when compiled

    def children(self): return []


class Lit(Expr):
    def __init__(self, value, rtype=None):
        if rtype is None:
            if isinstance(value, bool):
                rtype = K.T_BOOL
            elif isinstance(value, int):
                rtype = K.T_UINT64 if value >= 0 else K.T_INT64
            elif isinstance(value, float):
                rtype = K.T_FLOAT64
            elif isinstance(value, (str, bytes)):
                rtype = K.T_STRING
            else:
                raise TypeError("unsupported literal %r" % (value,))
        self.value = value
        self.rtype = rtype

    def children(self): return []

    def encode(self):
        """value bytes followed by the tag byte (VM stack element layout)"""
        t = self.rtype
        if t in (K.T_UINT64, K.T_TIMESTAMP64):
            return struct.pack("<QB", self.value & 0xFFFFFFFFFFFFFFFF, 0)
        if t == K.T_INT64:
            return struct.pack("<qB", self.value, 0)
        if t == K.T_FLOAT64:
            return struct.pack("<dB", float(self.value), 0)
        if t == K.T_BOOL:
            return struct.pack("<BB", 1 if self.value else 0, 0)
        if t == K.T_STRING:
            b = self.value.encode() if isinstance(self.value, str) else self.value
            return struct.pack("<I", len(b)) + b + b"\x00"
        raise TypeError("literal type")


class If(Expr):
    def __init__(self, cond, true_branch, false_branch):
        self.cond, self.t, self.f = _wrap(cond), _wrap(true_branch), _wrap(false_branch)

    def children(self): return [self.cond, self.t, self.f]


class Call(Expr):
    """pure function call; `name` is the SQL-level name (gt, add, to_int64 ...)"""

    def __init__(self, name, *args):
        self.name = name
        self.args = [_wrap(a) for a in args]

    def children(self): return self.args


class Agg(Expr):
    """aggregate call: count / sum / min / max / mean"""

    def __init__(self, name, arg):
        self.name = name
        self.arg = _wrap(arg)

    def children(self): return [self.arg]


def count(x=1): return Agg("count", x)
def sum_(x): return Agg("sum", x)
def min_(x): return Agg("min", x)
def max_(x): return Agg("max", x)
def mean(x): return Agg("mean", x)
def count_distinct(x): return Agg("count_distinct", x)
def col(name): return Col(name)
def lit(v, t=None): return Lit(v, t)


class CompileError(Exception):
    pass


def _columns_in(e, out):
    if isinstance(e, Col):
        if e.name not in out:
            out.append(e.name)
    for c in e.children():
        _columns_in(c, out)


def _find_agg(e):
    """first aggregate sub-expression, depth first (QueryTreeUtil.cc:209-224)"""
    if isinstance(e, Agg):
        return e
    for c in e.children():
        a = _find_agg(c)
        if a is not None:
            return a
    return None


class _Prog:
    def __init__(self):
        self.code = []      # (op, argt, arg0)
        self.static = b""

    def emit(self, op, arg0=0, argt=0):
        self.code.append([op, argt, arg0])
        return len(self.code) - 1


def _resolve_types(e, coltypes):
    """annotate rtype bottom-up; insert implicit conversions"""
    if isinstance(e, Col):
        if e.name not in coltypes:
            raise CompileError("column(s) not found: '%s'" % e.name)
        e.rtype = coltypes[e.name]
        return e
    if isinstance(e, Lit):
        return e
    if isinstance(e, If):
        e.cond = _resolve_types(e.cond, coltypes)
        e.t = _resolve_types(e.t, coltypes)
        e.f = _resolve_types(e.f, coltypes)
        if e.cond.rtype != K.T_BOOL:
            raise CompileError("type error: IF condition must be bool")
        if e.t.rtype != e.f.rtype:
            raise CompileError("type error: IF branches differ")
        e.rtype = e.t.rtype
        return e
    if isinstance(e, Agg):
        e.arg = _resolve_types(e.arg, coltypes)
        at = e.arg.rtype
        if e.name == "count":
            if at != K.T_NIL:
                e.arg = Call("to_nil", e.arg)
                e.arg.rtype = K.T_NIL
                e.arg.fn = K.FN(K.FAM_TO_NIL, _TS[at])
            e.fn, e.rtype = K.AGG_COUNT, K.T_UINT64
        elif e.name == "sum":
            m = {K.T_UINT64: (K.AGG_SUM_UINT64, K.T_UINT64),
                 K.T_INT64: (K.AGG_SUM_INT64, K.T_INT64),
                 K.T_FLOAT64: (K.AGG_SUM_FLOAT64, K.T_FLOAT64)}
            if at not in m:
                raise CompileError("type error for sum")
            e.fn, e.rtype = m[at]
        elif e.name in ("min", "max"):
            base = {K.T_UINT64: K.AGG_MIN_UINT64, K.T_INT64: K.AGG_MIN_INT64,
                    K.T_FLOAT64: K.AGG_MIN_FLOAT64}
            if at not in base:
                raise CompileError("type error for %s" % e.name)
            e.fn = base[at] + (1 if e.name == "max" else 0)
            e.rtype = at
        elif e.name == "mean":
            m = {K.T_UINT64: K.AGG_MEAN_UINT64, K.T_INT64: K.AGG_MEAN_INT64,
                 K.T_FLOAT64: K.AGG_MEAN_FLOAT64}
            if at not in m:
                raise CompileError("type error for mean")
            e.fn, e.rtype = m[at], K.T_FLOAT64
        elif e.name == "count_distinct":
            # count_distinct#uint64/uint64; (aggregate.cc:126-137)
            if at != K.T_UINT64:
                raise CompileError("type error for count_distinct")
            e.fn, e.rtype = K.AGG_COUNT_DISTINCT_UINT64, K.T_UINT64
        else:
            raise CompileError("method not found: %s" % e.name)
        return e
    if isinstance(e, Call):
        e.args = [_resolve_types(a, coltypes) for a in e.args]
        ats = [a.rtype for a in e.args]
        n = e.name
        if hasattr(e, "fn"):
            return e
        if n in ("logical_and", "logical_or"):
            if ats != [K.T_BOOL, K.T_BOOL]:
                raise CompileError("type error for %s" % n)
            e.fn = K.FN(K.FAM_LOGICAL_AND if n == "logical_and" else K.FAM_LOGICAL_OR, 0)
            e.rtype = K.T_BOOL
        elif n == "neg":
            if ats != [K.T_BOOL]:
                raise CompileError("type error for neg")
            e.fn, e.rtype = K.FN(K.FAM_NEG, 0), K.T_BOOL
        elif n == "add" and ats == [K.T_STRING, K.T_STRING]:
            # `add` is also registered with expressions::concat (defaults.cc: add -> concat)
            e.fn, e.rtype = K.FN(K.FAM_CONCAT, K.TS_STRING), K.T_STRING
        elif n in _CMP_FAMS or n in _ARITH_FAMS:
            if len(ats) != 2:
                raise CompileError("wrong number of arguments for %s" % n)
            # the only implicit numeric conversion: UINT64 -> INT64
            if ats[0] != ats[1]:
                if set(ats) == {K.T_UINT64, K.T_INT64}:
                    for i in (0, 1):
                        if ats[i] == K.T_UINT64:
                            c = Call("to_int64", e.args[i])
                            c.fn = K.FN(K.FAM_TO_INT64, K.TS_UINT64)
                            c.rtype = K.T_INT64
                            e.args[i] = c
                    ats = [K.T_INT64, K.T_INT64]
                else:
                    raise CompileError("type error for %s<%s>" % (n, ats))
            t = ats[0]
            if n in _CMP_FAMS:
                if t == K.T_BOOL and n not in ("eq", "neq"):
                    raise CompileError("type error for %s<bool>" % n)
                e.fn = K.FN(_CMP_FAMS[n], _TS[t])
                e.rtype = K.T_INT64 if n == "cmp" else K.T_BOOL
            else:
                if t not in (K.T_UINT64, K.T_INT64, K.T_FLOAT64):
                    raise CompileError("type error for %s" % n)
                e.fn = K.FN(_ARITH_FAMS[n], _TS[t])
                e.rtype = t
        elif n == "to_int64":
            e.fn, e.rtype = K.FN(K.FAM_TO_INT64, _TS[ats[0]]), K.T_INT64
        elif n == "to_nil":
            e.fn, e.rtype = K.FN(K.FAM_TO_NIL, _TS[ats[0]]), K.T_NIL
        elif n == "to_timestamp64":
            e.fn, e.rtype = K.FN(K.FAM_TO_TIMESTAMP64, _TS[ats[0]]), K.T_TIMESTAMP64
        elif n == "to_string":
            # conversion.cc:140-215; the timestamp overload shares to_string_uint64's body
            # and function pointer (the adapter reports it under the uint64 symbol)
            if len(ats) != 1 or ats[0] == K.T_STRING:
                raise CompileError("type error for to_string")
            slot = K.TS_UINT64 if ats[0] == K.T_TIMESTAMP64 else _TS[ats[0]]
            e.fn, e.rtype = K.FN(K.FAM_TO_STRING, slot), K.T_STRING
        elif n in ("concat",) or (n == "add" and ats == [K.T_STRING, K.T_STRING]):
            if ats != [K.T_STRING, K.T_STRING]:
                raise CompileError("type error for concat")
            e.fn, e.rtype = K.FN(K.FAM_CONCAT, K.TS_STRING), K.T_STRING
        elif n in _STR1_FAMS:
            if ats != [K.T_STRING]:
                raise CompileError("type error for %s" % n)
            e.fn, e.rtype = K.FN(_STR1_FAMS[n], K.TS_STRING), K.T_STRING
        elif n in ("startswith", "endswith"):
            if ats != [K.T_STRING, K.T_STRING]:
                raise CompileError("type error for %s" % n)
            e.fn = K.FN(K.FAM_STARTSWITH if n == "startswith" else K.FAM_ENDSWITH, K.TS_STRING)
            e.rtype = K.T_BOOL
        elif n in ("substring", "substr"):
            if len(ats) != 2 or ats[0] != K.T_STRING or ats[1] not in (K.T_INT64, K.T_UINT64):
                raise CompileError("type error for substring")
            if ats[1] == K.T_UINT64:  # implicit UINT64 -> INT64 (CallExpressionNode.cc:58-85)
                c = Call("to_int64", e.args[1])
                c.fn = K.FN(K.FAM_TO_INT64, K.TS_UINT64)
                c.rtype = K.T_INT64
                e.args[1] = c
            e.fn, e.rtype = K.FN(K.FAM_SUBSTRING, K.TS_STRING), K.T_STRING
        else:
            raise CompileError("method not found: %s" % n)
        return e
    raise CompileError("can't compile expression")


def _emit(e, p, colidx):
    if isinstance(e, Col):
        p.emit(K.X_INPUT, colidx[e.name], e.rtype)
    elif isinstance(e, Lit):
        off = len(p.static)
        p.static += e.encode()
        p.emit(K.X_LITERAL, off, e.rtype)
    elif isinstance(e, If):
        _emit(e.cond, p, colidx)
        j = p.emit(K.X_CJUMP, 0)
        _emit(e.f, p, colidx)
        p.code[j][2] = len(p.code) + 1
        j2 = p.emit(K.X_JUMP, 0)
        _emit(e.t, p, colidx)
        p.code[j2][2] = len(p.code)
    elif isinstance(e, Call):
        for a in e.args:
            _emit(a, p, colidx)
        p.emit(K.X_CALL_PURE, e.fn)
    elif isinstance(e, Agg):
        p.emit(K.X_CALL_INSTANCE, K.INSTANCE_GET)
    else:
        raise CompileError("can't compile expression")


class CompiledProgram:
    """owns the ctypes buffers of one evql_program_t"""

    def __init__(self, expr, coltypes, colidx):
        expr = _resolve_types(expr, coltypes)
        p = _Prog()
        _emit(expr, p, colidx)
        p.emit(K.X_RETURN)
        acc = 0
        aggfn = K.AGG_NONE
        a = _find_agg(expr)
        if a is not None:
            acc = len(p.code)
            _emit(a.arg, p, colidx)
            p.emit(K.X_CALL_INSTANCE, K.INSTANCE_ACCUMULATE)
            p.emit(K.X_RETURN)
            aggfn = a.fn
        self.expr = expr
        self.code = (K.Instr * len(p.code))(*[K.Instr(o, t, a0) for o, t, a0 in p.code])
        st = p.static if p.static else b"\x00"
        self.static = (C.c_uint8 * len(st)).from_buffer_copy(st)
        self.struct = K.Program(
            C.cast(self.code, C.POINTER(K.Instr)), len(p.code), 0, acc,
            expr.rtype, aggfn, C.cast(self.static, C.POINTER(C.c_uint8)),
            len(p.static))
        self.return_type = expr.rtype
        self.is_aggregate = acc > 0


_LOGICAL_TO_STYPE = {K.COL_BOOLEAN: K.T_BOOL, K.COL_UNSIGNED_INT: K.T_UINT64,
                     K.COL_SIGNED_INT: K.T_INT64, K.COL_STRING: K.T_STRING,
                     K.COL_FLOAT: K.T_FLOAT64, K.COL_DATETIME: K.T_TIMESTAMP64}


def stype_of_column(logical_type):
    return _LOGICAL_TO_STYPE[logical_type]


class Plan:
    """A GROUP BY over a sequential scan, lowered to an evql_plan_desc_t.

    schema: dict column name -> evql_stype
    where: Expr or None;  group_by: [Expr];  select: [Expr]
    A plan with neither group_by nor select is a bare scan of `scan_select`.
    """

    def __init__(self, schema, select=(), group_by=(), where=None,
                 scan_select=None, mode=K.MODE_FINAL, scan_mode=K.SCAN_FLAT,
                 groups_hint=0, row_filter=None, row_end=0, row_begin=0,
                 float_sum_mode=0, float_sum_bound=0.0):
        self.schema = dict(schema)
        # scan columns: WHERE first, then group exprs, then select exprs -- the
        # order in which QueryPlanBuilder resolves references
        names = []
        if where is not None:
            _columns_in(where, names)
        for e in list(group_by) + list(select) + list(scan_select or []):
            _columns_in(e, names)
        # out(i) ("$i") names output column i of an explicit scan select list
        names = [n for n in names if not n.startswith("$")]
        self.scan_columns = names
        coltypes = {n: self.schema[n] for n in names if n in self.schema}
        for n in names:
            if n not in self.schema:
                raise CompileError("column(s) not found: '%s'" % n)
        colidx = {n: i for i, n in enumerate(names)}

        self.where = CompiledProgram(where, coltypes, colidx) if where is not None else None
        if self.where is not None and self.where.return_type != K.T_BOOL:
            raise CompileError("WHERE must be a boolean expression")

        if scan_select is not None:
            # bare scan: explicit scan select list
            self.scan_select = [CompiledProgram(e, coltypes, colidx) for e in scan_select]
            # group / select expressions above an explicit scan select list (WITHIN
            # RECORD scans) reference its outputs as out(i)
            out_names = ["$%d" % i for i in range(len(self.scan_select))]
            coltypes = dict(coltypes)
            for i, p in enumerate(self.scan_select):
                coltypes["$%d" % i] = p.return_type
        else:
            # GROUP BY level: scan output = bare refs in first-use order
            out_names = []
            for e in list(group_by) + list(select):
                _columns_in(e, out_names)
            self.scan_select = [CompiledProgram(Col(n), coltypes, colidx) for n in out_names]
        out_idx = {n: i for i, n in enumerate(out_names)}
        out_types = {n: coltypes[n] for n in out_names}
        self.group = [CompiledProgram(e, out_types, out_idx) for e in group_by]
        self.select = [CompiledProgram(e, out_types, out_idx) for e in select]

        # ctypes assembly ---------------------------------------------------
        self._names = (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
        self._types = (C.c_uint32 * max(1, len(names)))(*[coltypes[n] for n in names])
        self._scan_select = (K.Program * max(1, len(self.scan_select)))(
            *[p.struct for p in self.scan_select])
        self._group = (K.Program * max(1, len(self.group)))(*[p.struct for p in self.group])
        self._select = (K.Program * max(1, len(self.select)))(*[p.struct for p in self.select])
        self._filter = None
        d = K.PlanDesc()
        d.scan_columns = C.cast(self._names, C.POINTER(C.c_char_p))
        d.scan_column_types = C.cast(self._types, C.POINTER(C.c_uint32))
        d.n_scan_columns = len(names)
        d.where = C.pointer(self.where.struct) if self.where is not None else None
        d.scan_select = C.cast(self._scan_select, C.POINTER(K.Program))
        d.n_scan_select = len(self.scan_select)
        d.group_exprs = C.cast(self._group, C.POINTER(K.Program))
        d.n_group = len(self.group)
        d.select_exprs = C.cast(self._select, C.POINTER(K.Program))
        d.n_select = len(self.select)
        if row_filter is not None:
            import numpy as np
            bits = np.packbits(np.asarray(row_filter, dtype=np.uint8), bitorder="little")
            self._filter = (C.c_uint8 * len(bits)).from_buffer_copy(bits.tobytes())
            d.row_filter_bits = C.cast(self._filter, C.POINTER(C.c_uint8))
            d.row_filter_len = len(row_filter)
        d.group_mode = mode
        d.scan_mode = scan_mode
        d.groups_hint = groups_hint
        d.row_begin = row_begin
        d.row_end = row_end
        d.float_sum_mode = float_sum_mode
        d.float_sum_bound = float_sum_bound
        self.desc = d

    @property
    def output_types(self):
        if self.select:
            return [p.return_type for p in self.select]
        return [p.return_type for p in self.scan_select]


def out(i):
    """output column i of a Plan, for ORDER BY expressions"""
    return Col("$%d" % i)


class Order:
    """ORDER BY specs (+ LIMIT / OFFSET) over a Plan's output columns, lowered to
    evql_sort_spec_t[] (OrderByExpression's sort_specs, orderby.cc:35-57).
    specs: [(expr over out(i) | int column index, descending)]"""

    def __init__(self, plan, specs=(), limit=None, offset=0):
        types = plan.output_types
        coltypes = {"$%d" % i: t for i, t in enumerate(types)}
        colidx = {"$%d" % i: i for i in range(len(types))}
        self.programs = []
        self.descending = []
        for e, desc in specs:
            if isinstance(e, int):
                e = out(e)
            self.programs.append(CompiledProgram(e, coltypes, colidx))
            self.descending.append(bool(desc))
        n = len(self.programs)
        self.specs = (K.SortSpec * max(1, n))(
            *[K.SortSpec(p.struct, int(d)) for p, d in zip(self.programs, self.descending)])
        self.n = n
        self.limit = -1 if limit is None else int(limit)
        self.offset = int(offset)


# ---------------------------------------------------------------------------
# decoding packed SVector bytes (sql/svalue.cc:410-517) into python values
# ---------------------------------------------------------------------------
def unpack_svector(stype, data):
    """bytes -> list of python values (None for STAG_NULL)"""
    out = []
    mv = memoryview(data)
    pos, n = 0, len(mv)
    if stype in (K.T_UINT64, K.T_TIMESTAMP64, K.T_INT64, K.T_FLOAT64):
        import numpy as np
        code = {K.T_UINT64: "<u8", K.T_TIMESTAMP64: "<u8", K.T_INT64: "<i8",
                K.T_FLOAT64: "<f8"}[stype]
        rec = np.frombuffer(data, dtype=np.dtype([("v", code), ("t", "u1")]))  # 9-byte elements
        out = rec["v"].tolist()
        nulls = np.nonzero(rec["t"] & K.STAG_NULL)[0]
        for i in nulls.tolist():
            out[i] = None
    elif stype == K.T_BOOL:
        while pos < n:
            out.append(None if mv[pos + 1] & K.STAG_NULL else bool(mv[pos]))
            pos += 2
    elif stype == K.T_STRING:
        while pos < n:
            l = struct.unpack_from("<I", mv, pos)[0]
            s = bytes(mv[pos + 4:pos + 4 + l])
            tag = mv[pos + 4 + l]
            out.append(None if tag & K.STAG_NULL else s)
            pos += 4 + l + 1
    elif stype == K.T_NIL:
        out = [None] * n
    return out


# ---------------------------------------------------------------------------
# plans from dumped bytecode
# ---------------------------------------------------------------------------
class RawProgram:
    """one evql_program_t from a dump {"code": [[op, argt, arg0, ...]], "static": hex,
    "method_call", "method_accumulate", "return_type", "aggregate_fn"} -- the form in
    which the reference-side adapter (eventql_amd/adapter/gpu_bridge.cc) lowers a
    real csql::vm::Program"""

    def __init__(self, d):
        code = d["code"]
        self.code = (K.Instr * max(1, len(code)))(*[K.Instr(c[0], c[1], c[2]) for c in code])
        st = bytes.fromhex(d["static"]) if d.get("static") else b""
        buf = st if st else b"\x00"
        self.static = (C.c_uint8 * len(buf)).from_buffer_copy(buf)
        self.struct = K.Program(
            C.cast(self.code, C.POINTER(K.Instr)), len(code), d["method_call"],
            d["method_accumulate"], d["return_type"], d["aggregate_fn"],
            C.cast(self.static, C.POINTER(C.c_uint8)), len(st))
        self.return_type = d["return_type"]
        self.is_aggregate = d["method_accumulate"] > 0

    @staticmethod
    def dump_of(compiled):
        """the same dict for a CompiledProgram (plan.py's own compiler)"""
        s = compiled.struct
        return dict(code=[[compiled.code[i].op, compiled.code[i].argt, compiled.code[i].arg0]
                          for i in range(s.code_len)],
                    static=bytes(compiled.static[:s.static_storage_len]).hex(),
                    method_call=s.method_call, method_accumulate=s.method_accumulate,
                    return_type=s.return_type, aggregate_fn=s.aggregate_fn)


class DumpedPlan:
    """evql_plan_desc_t assembled from dumped programs: {"scan_columns": [[name, stype]],
    "where": prog|None, "scan_select": [prog], "group": [prog], "select": [prog]}"""

    def __init__(self, dump, mode=K.MODE_FINAL, scan_mode=K.SCAN_FLAT, groups_hint=0):
        names = [c[0] for c in dump["scan_columns"]]
        self.scan_columns = names
        self.where = RawProgram(dump["where"]) if dump.get("where") else None
        self.scan_select = [RawProgram(p) for p in dump["scan_select"]]
        self.group = [RawProgram(p) for p in dump["group"]]
        self.select = [RawProgram(p) for p in dump["select"]]
        self._names = (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
        self._types = (C.c_uint32 * max(1, len(names)))(*[c[1] for c in dump["scan_columns"]])
        self._scan_select = (K.Program * max(1, len(self.scan_select)))(
            *[p.struct for p in self.scan_select])
        self._group = (K.Program * max(1, len(self.group)))(*[p.struct for p in self.group])
        self._select = (K.Program * max(1, len(self.select)))(*[p.struct for p in self.select])
        d = K.PlanDesc()
        d.scan_columns = C.cast(self._names, C.POINTER(C.c_char_p))
        d.scan_column_types = C.cast(self._types, C.POINTER(C.c_uint32))
        d.n_scan_columns = len(names)
        d.where = C.pointer(self.where.struct) if self.where is not None else None
        d.scan_select = C.cast(self._scan_select, C.POINTER(K.Program))
        d.n_scan_select = len(self.scan_select)
        d.group_exprs = C.cast(self._group, C.POINTER(K.Program))
        d.n_group = len(self.group)
        d.select_exprs = C.cast(self._select, C.POINTER(K.Program))
        d.n_select = len(self.select)
        d.group_mode = mode
        d.scan_mode = scan_mode
        d.groups_hint = groups_hint
        self.desc = d

    @property
    def output_types(self):
        if self.select:
            return [p.return_type for p in self.select]
        return [p.return_type for p in self.scan_select]


def dump_of_plan(plan):
    """a Plan's programs in the dump form, for comparison with the reference's"""
    return dict(
        scan_columns=[[n, int(plan._types[i])] for i, n in enumerate(plan.scan_columns)],
        where=RawProgram.dump_of(plan.where) if plan.where is not None else None,
        scan_select=[RawProgram.dump_of(p) for p in plan.scan_select],
        group=[RawProgram.dump_of(p) for p in plan.group],
        select=[RawProgram.dump_of(p) for p in plan.select])
