"""Plan expression trees <-> SQL text the reference's parser accepts.

The golden fixtures tests/golden/ref_csql_*.json are produced by running SQL through
the REAL reference engine (oracle/_ref/csql_probe, see tests/golden/gen_ref_csql.py).
The cases are generated as `eventql_amd.plan` expression trees from a seed and
rendered to SQL here, so that a test can rebuild the very same plan on a box that has
no reference tree.

Reference behaviour this module has to respect (all observed by running it):
  * QueryPlanBuilder folds constant sub-expressions into literals
    (sql/runtime/queryplanbuilder.cc:1395-1407, qtree/QueryTreeUtil.cc:46-80):
    `fold()` does the same on the expression tree with the reference's arithmetic;
  * a column that the select list uses must also appear in WHERE or GROUP BY, or the
    planner fails with "column(s) not found" (queryplanbuilder.cc:501-512):
    `make_runnable()` adds a tautological conjunct per such column;
  * `-x` parses as neg(x) (boolean NOT) and is a type error on numbers: negative
    literals are written `(0.0 - 2.25)`, which the planner folds;
  * NOT binds more loosely than AND/OR in the reference's parser: everything is
    parenthesised.
"""
import math
import struct

from eventql_amd import capi as K
from eventql_amd.plan import Col, Lit, Call, If, Agg

_INFIX = {"add": "+", "sub": "-", "mul": "*", "div": "/", "mod": "%", "eq": "=",
          "neq": "!=", "lt": "<", "lte": "<=", "gt": ">", "gte": ">=",
          "logical_and": "AND", "logical_or": "OR"}
M64 = (1 << 64) - 1


class NotRenderable(Exception):
    pass


def render(e):
    if isinstance(e, Col):
        return e.name
    if isinstance(e, Lit):
        t, v = e.rtype, e.value
        if t == K.T_BOOL:
            return "true" if v else "false"
        if t == K.T_UINT64:
            return "%d" % v
        if t == K.T_INT64 and v < 0:  # "-5" is an INT64 literal (queryplanbuilder.cc:1519-1529)
            return "%d" % v
        if t == K.T_FLOAT64:
            if v != v or v in (float("inf"), float("-inf")):
                raise NotRenderable("non-finite float literal")
            if v < 0 or (v == 0 and math.copysign(1, v) < 0):
                return "(0.0 - %s)" % _float_text(-v)
            return _float_text(v)
        if t == K.T_STRING:
            s = v.decode() if isinstance(v, bytes) else v
            if "'" in s or "\\" in s:
                raise NotRenderable("quote in string literal")
            return "'%s'" % s
        raise NotRenderable("literal type %r" % t)
    if isinstance(e, If):
        return "if(%s, %s, %s)" % (render(e.cond), render(e.t), render(e.f))
    if isinstance(e, Agg):
        if e.name not in ("count", "sum", "count_distinct"):
            raise NotRenderable("aggregate %s does not exist in the reference" % e.name)
        return "%s(%s)" % (e.name, render(e.arg))
    if isinstance(e, Call):
        if e.name == "neg":
            return "(NOT (%s))" % render(e.args[0])
        if e.name == "pow":
            return "pow(%s, %s)" % (render(e.args[0]), render(e.args[1]))
        if e.name in _INFIX:
            return "(%s %s %s)" % (render(e.args[0]), _INFIX[e.name], render(e.args[1]))
        if e.name in ("to_string", "concat", "lcase", "ucase", "lowercase", "uppercase", "ltrim",
                      "rtrim", "substring", "substr", "startswith", "endswith"):
            return "%s(%s)" % (e.name, ", ".join(render(a) for a in e.args))
        raise NotRenderable("call %s" % e.name)
    raise NotRenderable(repr(e))


def _float_text(v):
    s = repr(float(v))
    if "e" in s or "E" in s or "." not in s:
        s = "%.17f" % v
        if float(s) != v:
            raise NotRenderable("float literal not exactly printable")
    return s


# --------------------------------------------------------------------------------------
# constant folding with the reference's arithmetic (math.cc / boolean.cc)
# --------------------------------------------------------------------------------------
def _is_const(e):
    if isinstance(e, Col) or isinstance(e, Agg):
        return False
    if isinstance(e, Lit):
        return True
    if isinstance(e, If):
        return all(_is_const(c) for c in (e.cond, e.t, e.f))
    return all(_is_const(a) for a in e.args)


class FoldError(Exception):
    """the reference raises while folding (e.g. division by zero at plan time)"""


def _lit_type(e):
    return e.rtype


def _eval(e):
    if isinstance(e, Lit):
        return e.rtype, e.value
    if isinstance(e, If):
        ct, cv = _eval(e.cond)
        # the VM evaluates only the branch taken (compiler.cc:174-209)
        return _eval(e.t if cv else e.f)
    n = e.name
    vals = [_eval(a) for a in e.args]
    if n in ("logical_and", "logical_or"):
        a, b = bool(vals[0][1]), bool(vals[1][1])
        return K.T_BOOL, (a and b) if n == "logical_and" else (a or b)
    if n == "neg":
        return K.T_BOOL, not bool(vals[0][1])
    if len(vals) != 2:
        raise FoldError("cannot fold %s" % n)
    (ta, a), (tb, b) = vals
    if ta != tb:
        raise FoldError("mixed literal types")
    if n in ("eq", "neq", "lt", "lte", "gt", "gte"):
        r = {"eq": a == b, "neq": a != b, "lt": a < b, "lte": a <= b, "gt": a > b,
             "gte": a >= b}[n]
        return K.T_BOOL, r
    if ta == K.T_UINT64:
        if n == "add":
            return ta, (a + b) & M64
        if n == "sub":
            return ta, (a - b) & M64
        if n == "mul":
            return ta, (a * b) & M64
        if n in ("div", "mod"):
            if b == 0:
                raise FoldError("division by zero")
            return ta, (a // b) if n == "div" else (a % b)
    if ta == K.T_FLOAT64:
        if n == "add":
            return ta, a + b
        if n == "sub":
            return ta, a - b
        if n == "mul":
            return ta, a * b
        if n == "div":
            if b == 0:
                raise FoldError("float division by zero (inf/nan literal)")
            return ta, a / b
    raise FoldError("cannot fold %s" % n)


def fold(e):
    """bottom-up: every column-free subtree becomes one literal"""
    if isinstance(e, (Col, Lit)):
        return e
    if isinstance(e, Agg):
        return Agg(e.name, fold(e.arg))
    if _is_const(e):
        t, v = _eval(e)
        if t == K.T_FLOAT64 and (v != v or v in (float("inf"), float("-inf"))):
            raise FoldError("non-finite folded literal")
        return Lit(v, t)
    if isinstance(e, If):
        return If(fold(e.cond), fold(e.t), fold(e.f))
    return Call(e.name, *[fold(a) for a in e.args])


def _columns(e, out):
    if isinstance(e, Col):
        if e.name not in out:
            out.append(e.name)
        return
    if isinstance(e, Lit):
        return
    for c in e.children():
        _columns(c, out)


def _tautology(name, stype):
    c = Col(name)
    if stype in (K.T_UINT64, K.T_TIMESTAMP64):
        # NULL compares as 0 (vm.cc:231-272): true for every row
        return Call("gte", c, Lit(0 if stype == K.T_UINT64 else 0, stype))
    if stype == K.T_FLOAT64:
        return Call("logical_or", Call("gte", c, Lit(0.0)), Call("lt", c, Lit(0.0)))
    if stype == K.T_BOOL:
        return Call("logical_or", Call("eq", c, Lit(True)), Call("eq", c, Lit(False)))
    if stype == K.T_STRING:
        return Call("logical_or", Call("eq", c, Lit("")), Call("neq", c, Lit("")))
    raise NotRenderable("column type")


def make_runnable(kw, schema, scan_order=False):
    """folds constants and makes every select-list column visible to the reference's
    planner.  Returns a new kwargs dict (same keys as Plan takes).

    scan_order: for the Dremel CSTableScan.  That operator DECLARES its output columns
    from SequentialScanNode::selectedColumns() -- the input columns in the order the
    planner first met them, WHERE first (CSTableScan.cc:53-56, 631-638) -- but FILLS
    them in select-list order (:471-486).  Unless both orders agree the parent pops
    values of the wrong width off the VM stack (observed: `select clicked, time ..
    where time >= 0` returns types [uint64, bool] and stale stack bytes).  With
    scan_order the WHERE starts with one tautology per scan output column, in select
    list order, so that the reference is in its well-defined regime; a predicate over
    other columns is not expressible then."""
    kw = dict(kw)
    kw["select"] = [fold(e) for e in kw.get("select", [])]
    kw["group_by"] = [fold(e) for e in kw.get("group_by", [])]
    where = fold(kw["where"]) if kw.get("where") is not None else None
    if scan_order:
        order = []
        for e in kw["group_by"] + kw["select"]:
            _columns(e, order)
        wcols = []
        if where is not None:
            _columns(where, wcols)
        if not set(wcols) <= set(order):
            raise NotRenderable("WHERE over columns outside the scan's select list")
        for name in reversed(order):
            t = _tautology(name, schema[name])
            where = t if where is None else Call("logical_and", t, where)
        kw["where"] = where
        return kw
    seen = []
    if where is not None:
        _columns(where, seen)
    for e in kw["group_by"]:
        _columns(e, seen)
    needed = []
    for e in kw["select"]:
        _columns(e, needed)
    for name in needed:
        if name in seen:
            continue
        t = _tautology(name, schema[name])
        where = t if where is None else Call("logical_and", where, t)
        seen.append(name)
    if where is not None:
        kw["where"] = where
    else:
        kw.pop("where", None)
    return kw


def sql_of(kw, table="t"):
    """SELECT text of a (made-runnable) plan"""
    if kw.get("row_filter") is not None or kw.get("row_end") or kw.get("row_begin"):
        raise NotRenderable("row filters / ranges are API-level, not SQL")
    sel = ", ".join(render(e) for e in kw["select"])
    s = "select %s from %s" % (sel, table)
    if kw.get("where") is not None:
        s += " where %s" % render(kw["where"])
    if kw.get("group_by"):
        s += " group by %s" % ", ".join(render(e) for e in kw["group_by"])
    return s + ";"


# --------------------------------------------------------------------------------------
# canonical form of a result, shared by the generator and the tests
# --------------------------------------------------------------------------------------
def canon_cell(stype, v):
    """python value -> JSON-able canonical cell: floats as their IEEE bits"""
    if v is None:
        return None
    if stype == K.T_FLOAT64:
        return "f:%016x" % struct.unpack("<Q", struct.pack("<d", v))[0]
    if stype == K.T_BOOL:
        return bool(v)
    if stype == K.T_STRING:
        return v.decode("latin-1") if isinstance(v, bytes) else v
    return int(v)


def canon_rows(types, rows):
    def key(r):
        return [(0, "") if c is None else (1, repr(c)) for c in r]
    out = [[canon_cell(t, c) for t, c in zip(types, r)] for r in rows]
    out.sort(key=key)
    return out


def rows_digest(rows):
    import hashlib
    import json
    return hashlib.sha1(json.dumps(rows, separators=(",", ":")).encode()).hexdigest()
