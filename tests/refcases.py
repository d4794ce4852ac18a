"""The cases of the reference-generated golden fixtures tests/golden/ref_csql_*.json.

Every case is a plan (expression trees, built from a seed or written out here) that
tests/sqlgen.py renders to SQL.  tests/golden/gen_ref_csql.py ran that SQL through the
REAL reference engine (oracle/_ref/csql_probe) in this container and committed what
came back: result rows, the compiled vm::Programs of the GROUP BY and of the scan
(lowered by the reference-side adapter), and the PartialGroupByExpression rows.  The
tests rebuild the same plans from this module, so nothing of the reference is needed
where they run.
"""
import random

import numpy as np

from eventql_amd import capi as K
from eventql_amd.plan import Col, Lit, Call, If, Agg
import sqlgen
import tables as T
from test_gpu_fuzz import Gen, NestedGen


class RefGen(Gen):
    """Gen restricted to what this snapshot of the reference implements: count,
    count_distinct(uint64), sum(uint64) (sql/defaults.cc:49-54); no float sums,
    min, max or mean; count(<string>) is a type error in the reference
    (conversion.cc:76-79 declares to_nil_string with a BOOL argument)"""
    count_cols = ()

    def aggregate(self):
        r = self.r
        c = r.random()
        if c < 0.25:
            return Agg("count", Lit(1))
        if c < 0.4:
            return Agg("count", Col(r.choice(self.count_cols or self.uint_cols)))
        if c < 0.5:
            return Agg("count_distinct", self.uint())
        if c < 0.9:
            return Agg("sum", self.uint())
        # post-aggregate arithmetic and IF inside the argument
        if r.random() < 0.5:
            return Call("add", Agg("sum", self.uint()), Lit(r.choice(self.lits)))
        return Agg("sum", If(self.boolean(1), Lit(1), Lit(0)))

    def plan_kwargs(self, row_ends):
        kw = Gen.plan_kwargs(self, row_ends)
        kw.pop("row_end", None)      # API-level (setFilter / partition slices), not SQL
        kw.pop("row_filter", None)
        return kw


class RefNestedGen(NestedGen, RefGen):
    aggregate = RefGen.aggregate
    schema = None

    def plan_kwargs(self, row_ends):
        kw = NestedGen.plan_kwargs(self, row_ends)
        kw.pop("row_end", None)
        kw.pop("row_filter", None)
        if "where" not in kw and self.r.random() < 0.7:
            # a predicate over columns of DIFFERENT repetition depth: after a rejected
            # row the reference resets parent values without re-reading them
            # (CSTableScan.cc:501-512).  Only over columns the plan selects anyway, so
            # that the scan stays in its well-defined regime (sqlgen.make_runnable)
            used = []
            for e in kw["group_by"] + kw["select"]:
                sqlgen._columns(e, used)
            u = [c for c in used if self.schema[c] == K.T_UINT64]
            b = [c for c in used if self.schema[c] == K.T_BOOL]
            if u:
                g2 = RefGen(self.r.randrange(1 << 30), uint_cols=u, float_cols=[], bool_cols=b,
                            key_cols=u, first_cols=u, lits=self.lits)
                g2.flt = lambda depth=0: Lit(g2.r.choice([0.0, 1.5, 100.0]))
                kw["where"] = g2.boolean()
        return kw


class RefStrGen(RefGen):
    """string-PRODUCING expressions in the select list (expressions/string.cc,
    conversion.cc:140-215, math.cc add#string): random trees over the string columns,
    conversions of integer / float / bool expressions and of aggregates"""
    str_cols = ("s", "ns")
    str_lits = ("", "x", "G", "g1", " pad  ", "Key-", "7")

    def strx(self, depth=0):
        r = self.r
        c = r.random()
        if depth > 2 or c < 0.3:
            return Col(r.choice(self.str_cols))
        if c < 0.5:
            # (a literal on one side only: the reference folds constant calls)
            a, b = self.strx(depth + 1), Lit(r.choice(self.str_lits))
            return Call("concat", a, b) if r.random() < 0.5 else Call("concat", b, a)
        if c < 0.6:
            return Call("concat", self.strx(depth + 1), self.strx(depth + 1))
        if c < 0.75:
            return Call(r.choice(["lcase", "ucase", "ltrim", "rtrim"]), self.strx(depth + 1))
        if c < 0.9:
            if r.random() < 0.6:
                pos = Lit(r.choice([0, 1, 2, 3, 5, 9, 40]))
            elif r.random() < 0.5:
                pos = Call("mod", Col(r.choice(self.uint_cols)), Lit(r.choice([3, 6])))
            else:
                # wraps below zero: to_int64 makes it a position counted from the end
                pos = Call("sub", Call("mod", Col(r.choice(self.uint_cols)), Lit(3)), Lit(r.choice([2, 4])))
            return Call("substring", self.strx(depth + 1), pos)
        if c < 0.96:
            return Call("to_string", self.stringable(depth + 1))
        return If(self.boolean(2), self.strx(depth + 1), self.strx(depth + 1))

    def stringable(self, depth=0):
        r = self.r
        c = r.random()
        if c < 0.5:
            return self.uint(depth + 1)
        if c < 0.7:
            return Col(r.choice(self.float_cols))
        if c < 0.85:
            return self.boolean(2)
        return Col("t")

    def plan_kwargs(self, row_ends):
        r = self.r
        c = r.random()
        if c < 0.55:
            keys = [Col(r.choice(["k", "k10", "nb"]))]
        elif c < 0.85:
            keys = [Col(r.choice(["s", "ns"]))]
        elif c < 0.95:
            keys = [Col("k10"), Col("ns")]
        else:
            keys = []
        select = list(keys)
        for _ in range(r.randint(1, 3)):
            c = r.random()
            if c < 0.7:
                select.append(self.strx())
            elif c < 0.85:
                select.append(Call(r.choice(["startswith", "endswith"]), self.strx(1),
                                   Lit(r.choice(self.str_lits))))
            else:
                select.append(Call("concat", Lit("n="), Call("to_string", self.aggregate())))
        select.append(self.aggregate())
        kw = dict(select=select, group_by=keys)
        # (few groups: every string value of the result is compared)
        kw["where"] = Call("logical_and", Call("lt", Col("k"), Lit(r.choice([8, 30, 120]))),
                           self.boolean(1)) if r.random() < 0.6 else Call("lt", Col("k"), Lit(60))
        kw["groups_hint"] = r.choice([0, 0, 1000])
        return kw


MIXED = dict(uint_cols=["k", "a", "b", "n", "p", "k10", "nb", "w"], float_cols=["v", "nv"],
             bool_cols=["f"], key_cols=["k", "k10", "f", "nb", "n", "s", "ns", "b"],
             first_cols=["a", "v", "s"], lits=[0, 1, 2, 7, 1000, 30000, 65535, 1 << 40])
MIXED_COUNT_COLS = ["k", "a", "n", "nb", "v", "nv", "f", "t"]

SUITES = {}


def suite(name):
    def deco(fn):
        SUITES[name] = fn
        return fn
    return deco


def partition_files(key):
    """"lsm:<name>" -> [(file name, image, has_skiplist, has_updates, columns)], oldest
    first (tests/lsm_tables.py)"""
    import lsm_tables
    assert key.startswith("lsm:")
    return lsm_tables.partition(key[4:])


def table_image(key):
    """(cstable image bytes, schema, probe scan kind) of a fixture table"""
    import nested_tables as N
    if key == "mixed":
        return T.mixed_table(300_000)[0], T.MIXED_SCHEMA, "fast"
    if key == "ranges":
        return T.ranges_table(), T.RANGES_SCHEMA, "fast"
    if key == "survey":
        return T.survey_table(1_000_000)[0], T.SURVEY_SCHEMA, "fast"
    if key == "items":
        return N.items_table(50_000)[0], N.ITEMS_SCHEMA, "dremel"
    if key == "testtbl":
        # the reference's own checked-in fixture, read as the v0.1.0 file it is
        return open(T.GOLDEN + "/testtbl.cst", "rb").read(), N.SIBLING_SCHEMA, "dremel"
    raise KeyError(key)


def _case(cid, table, kw, schema, scan_mode=K.SCAN_FLAT):
    """None when the plan cannot be expressed in the reference's SQL"""
    try:
        kw = sqlgen.make_runnable(kw, schema, scan_order=(scan_mode != K.SCAN_FLAT))
        sql = sqlgen.sql_of(kw, "t")
    except (sqlgen.NotRenderable, sqlgen.FoldError):
        return None
    return dict(id=cid, table=table, kw=kw, sql=sql, scan_mode=scan_mode)


@suite("mixed")
def mixed_cases():
    _, schema, _ = None, T.MIXED_SCHEMA, None
    out = []
    for seed in range(120):
        g = RefGen(50_000 + seed, **MIXED)
        g.count_cols = MIXED_COUNT_COLS
        c = _case("mixed-%03d" % seed, "mixed", g.plan_kwargs([1]), T.MIXED_SCHEMA)
        if c:
            out.append(c)
    return out


@suite("strings")
def strings_cases():
    out = []
    for seed in range(80):
        g = RefStrGen(90_000 + seed, **MIXED)
        g.count_cols = MIXED_COUNT_COLS
        c = _case("strings-%03d" % seed, "mixed", g.plan_kwargs([1]), T.MIXED_SCHEMA)
        if c:
            out.append(c)
    # startswith / endswith as predicates and keys: evaluated per row on the device
    k10, a, s, ns = Col("k10"), Col("a"), Col("s"), Col("ns")
    affix = [
        dict(select=[k10, Agg("count", Lit(1)), Agg("sum", a)], group_by=[k10],
             where=Call("startswith", s, Lit("g1"))),
        dict(select=[k10, Agg("count", Lit(1))], group_by=[k10], where=Call("endswith", s, Lit("7"))),
        dict(select=[Agg("count", Lit(1)), Agg("sum", a)], group_by=[],
             where=Call("logical_and", Call("startswith", ns, Lit("s4")),
                        Call("neg", Call("endswith", ns, Lit("9"))))),
        dict(select=[Agg("count", Lit(1))], group_by=[], where=Call("startswith", s, Lit(""))),
        dict(select=[Agg("count", Lit(1))], group_by=[], where=Call("endswith", ns, Lit(""))),
        dict(select=[Agg("count", Lit(1))], group_by=[],
             where=Call("startswith", s, Lit("g123456789012345678901234567890"))),
        dict(select=[Call("startswith", s, Lit("g9")), Agg("count", Lit(1)), Agg("sum", a)],
             group_by=[Call("startswith", s, Lit("g9"))]),
        dict(select=[Call("endswith", ns, Lit("0")), k10, Agg("count", Lit(1))],
             group_by=[Call("endswith", ns, Lit("0")), k10], where=Call("lt", Col("k"), Lit(300))),
        dict(select=[s, Agg("count", Lit(1))], group_by=[s],
             where=Call("logical_or", Call("endswith", s, Lit("99")), Call("startswith", s, s))),
        dict(select=[k10, Agg("sum", If(Call("startswith", s, Lit("g2")), a, Lit(0)))], group_by=[k10],
             where=Call("gte", a, Lit(0))),
    ]
    for i, kw in enumerate(affix):
        c = _case("strings-affix-%02d" % i, "mixed", kw, T.MIXED_SCHEMA)
        assert c is not None, i
        out.append(c)
    # no GROUP BY, non-aggregate select expressions: the ONE group's values come from the
    # first row that passes WHERE (groupby.cc:161-172)
    v, p, k = Col("v"), Col("p"), Col("k")
    glob = [
        dict(select=[ns, Call("ucase", s), p, Agg("count", Lit(1))], group_by=[],
             where=Call("logical_and", Call("lt", k, Lit(60)), Call("gt", a, Lit(1000)))),
        dict(select=[Call("concat", ns, Lit("x")), Agg("sum", a)], group_by=[],
             where=Call("gt", a, Lit(65000))),
        dict(select=[s, Agg("count", Lit(1))], group_by=[], where=Call("gt", v, Lit(9.0e9))),
        # (`p + count(1)` -- an input beside the aggregate -- trips an assertion in the
        # reference, vm.cc:139: not pinnable)
        dict(select=[p, v, Call("add", Agg("count", Lit(1)), Lit(7))], group_by=[],
             where=Call("eq", Call("mod", a, Lit(9973)), Lit(17))),
        dict(select=[ns, Agg("count_distinct", k)], group_by=[],
             where=Call("logical_and", Call("lt", k, Lit(30)), Call("lte", Lit(1.5), v))),
    ]
    for i, kw in enumerate(glob):
        c = _case("strings-global-%02d" % i, "mixed", kw, T.MIXED_SCHEMA)
        assert c is not None, i
        out.append(c)
    # IF over strings (compiler.cc:174-209: only the branch taken is evaluated)
    one = Agg("count", Lit(1))
    ifs = [
        dict(select=[k10, If(Call("gt", a, Lit(30000)), s, Lit("x")), one], group_by=[k10]),
        dict(select=[k10, Call("concat", If(Call("gt", a, Lit(30000)), s, ns), Lit("!")), one],
             group_by=[k10], where=Call("lt", k, Lit(200))),
        dict(select=[k10, If(Call("gt", one, Lit(29990)), Lit("many"), Lit("few")), one], group_by=[k10]),
        dict(select=[s, If(Call("startswith", s, Lit("g1")), Call("ucase", s), Call("to_string", a)), one],
             group_by=[s], where=Call("lt", k, Lit(40))),
        dict(select=[k10, Call("substring", If(Col("f"), ns, s), Lit(2)), one], group_by=[k10]),
    ]
    for i, kw in enumerate(ifs):
        c = _case("strings-if-%02d" % i, "mixed", kw, T.MIXED_SCHEMA)
        assert c is not None, i
        out.append(c)
    return out


@suite("ranges")
def ranges_cases():
    out = []
    for seed in range(60):
        g = RefGen(60_000 + seed, **T.RANGES)
        c = _case("ranges-%03d" % seed, "ranges", g.plan_kwargs([1]), T.RANGES_SCHEMA)
        if c:
            out.append(c)
    return out


def _c(name):
    return Col(name)


@suite("survey")
def survey_cases():
    """BASELINE.json's configs on the 1M-row SURVEY 8c(ii) table, and every scalar of
    tests/golden/survey_8c.json as a regenerable case"""
    a, b, k, n, s, v = [_c(x) for x in "abknsv"]
    f3 = Call("logical_and", Call("gt", a, Lit(30000)), Call("lt", b, Lit(30000)))
    cases = [
        ("config1-count", dict(select=[Agg("count", Lit(1))], group_by=[])),
        ("config2-int-twin", dict(select=[k, Agg("sum", a), Agg("count", Lit(1))], group_by=[k])),
        ("config3", dict(select=[k, Agg("sum", a), Agg("count", Lit(1)), Agg("sum", b)],
                         group_by=[k], where=f3)),
        ("config3-count-only", dict(select=[Agg("count", Lit(1))], group_by=[], where=f3)),
        ("config4-string-keys", dict(select=[s, Agg("count", Lit(1)), Agg("sum", a), Agg("sum", b)],
                                     group_by=[s])),
        ("config4-high-card", dict(select=[Call("mod", Call("mul", a, b), Lit(666667)),
                                           Agg("count", Lit(1)), Agg("sum", a)],
                                   group_by=[Call("mod", Call("mul", a, b), Lit(666667))])),
        ("null-sum-count", dict(select=[Agg("sum", n), Agg("count", n)], group_by=[],
                                where=Call("gte", n, Lit(0)))),
        ("null-compares-as-zero", dict(select=[Agg("count", Lit(1))], group_by=[],
                                       where=Call("gt", n, Lit(5)))),
        ("not-or", dict(select=[Agg("count", Lit(1))], group_by=[],
                        where=Call("logical_or", Call("neg", Call("gt", a, Lit(30000))),
                                   Call("eq", b, Lit(5))))),
        ("null-group", dict(select=[n, Agg("count", Lit(1))], group_by=[n],
                            where=Call("lt", n, Lit(10)))),
        ("uint64-wrap", dict(select=[Agg("sum", Call("mul", b, Lit(1 << 48)))], group_by=[],
                             where=Call("gte", b, Lit(0)))),
        ("sum-if", dict(select=[k, Agg("sum", If(Call("gt", a, Lit(30000)), Lit(1), Lit(0)))],
                        group_by=[k], where=Call("gte", a, Lit(0)))),
        ("one-instance-quirk", dict(select=[k, Call("add", Agg("sum", a), Agg("sum", b))],
                                    group_by=[k],
                                    where=Call("logical_and", Call("gte", a, Lit(0)),
                                               Call("gte", b, Lit(0))))),
        ("multi-key", dict(select=[k, s, Agg("count", Lit(1)), Agg("sum", a)], group_by=[k, s],
                           where=Call("gte", a, Lit(0)))),
        ("first-row", dict(select=[k, a, v, s, Agg("count", Lit(1))], group_by=[k],
                           where=Call("logical_and", Call("gte", a, Lit(0)),
                                      Call("logical_and", Call("gte", v, Lit(0.0)),
                                           Call("neq", s, Lit("")))))),
        ("zero-rows", dict(select=[Agg("count", Lit(1))], group_by=[],
                           where=Call("gt", v, Lit(8000000.5)))),
        ("count-distinct", dict(select=[k, Agg("count_distinct", a)], group_by=[k],
                                where=Call("gte", a, Lit(0)))),
        ("float-predicate", dict(select=[Call("mod", k, Lit(7)), Agg("count", Lit(1))],
                                 group_by=[Call("mod", k, Lit(7))],
                                 where=Call("lt", Call("mul", v, Lit(2.0)), Lit(8000.5)))),
        ("string-predicate", dict(select=[Agg("count", Lit(1)), Agg("sum", a)], group_by=[],
                                  where=Call("logical_and", Call("gte", s, Lit("g5")),
                                             Call("gte", a, Lit(0))))),
        # string-PRODUCING functions in the select list (expressions/string.cc,
        # conversion.cc:140-215): evaluated once per group over its first row
        ("string-functions", dict(
            select=[k, Call("concat", s, Lit("/x")), Call("add", Lit("k="), Call("to_string", k)),
                    Call("ucase", s), Call("substring", s, Lit(2)),
                    Call("substring", s, Lit(9)), Call("startswith", s, Lit("g1")),
                    Call("endswith", s, Lit("7")), Call("to_string", v), Agg("count", Lit(1))],
            group_by=[k], where=Call("gt", a, Lit(65000)))),
        ("string-of-aggregates", dict(
            select=[k, Call("concat", Lit("n="), Call("to_string", Agg("sum", a))),
                    Call("to_string", n), Call("ltrim", Call("concat", Lit("  "), s)),
                    Call("rtrim", Call("concat", s, Lit(" y  "))),
                    Call("lcase", Call("concat", Lit("MiXeD"), s)),
                    Call("to_string", Call("gt", a, Lit(40000)))],
            group_by=[k], where=Call("lt", k, Lit(40)))),
        ("string-keyed-functions", dict(
            select=[s, Call("concat", s, s), Call("to_string", Call("add", Agg("count", Lit(1)), Lit(1)))],
            group_by=[s], where=Call("lt", k, Lit(25)))),
    ]
    out = []
    for cid, kw in cases:
        c = _case("survey-" + cid, "survey", kw, T.SURVEY_SCHEMA)
        assert c is not None, cid
        out.append(c)
    return out


ITEMS = dict(uint_cols=["id", "items.position", "items.price"], float_cols=["score"],
             bool_cols=[], key_cols=["items.position", "id"], first_cols=["items.price", "id"],
             lits=[0, 1, 3, 7, 1000, 50000, 1 << 33])
TESTTBL = dict(uint_cols=["time", "event.search_query.time",
                          "event.search_query.num_result_items",
                          "event.search_query.result_items.position"],
               float_cols=[], bool_cols=["event.search_query.result_items.clicked"],
               key_cols=["event.search_query.result_items.position",
                         "event.search_query.num_result_items",
                         "event.search_query.result_items.clicked",
                         # string fields of every repetition depth (0, 1, 2)
                         "session_id", "event.search_query.query_string",
                         "event.search_query.result_items.item_id"],
               first_cols=["time", "event.search_query.num_result_items",
                           "event.search_query.query_string", "session_id"],
               lits=[0, 1, 2, 6, 10, 1438055327])


@suite("nested")
def nested_cases():
    """CSTableScan (Dremel assembly), AggregationStrategy::NO_AGGREGATION"""
    import nested_tables as N
    out = []
    for seed in range(120):
        which = seed % 2
        table, schema, cols = (("items", N.ITEMS_SCHEMA, ITEMS) if which == 0 else
                               ("testtbl", N.NESTED_SCHEMA, TESTTBL))
        g = RefNestedGen(70_000 + seed, **cols)
        g.schema = schema
        g.count_cols = list(cols["uint_cols"]) + list(cols["bool_cols"])
        if which == 0:
            g.leaf_uint, g.leaf_bool = ["items.position", "items.price"], []
        else:
            g.leaf_uint = ["event.search_query.result_items.position"]
            g.leaf_bool = ["event.search_query.result_items.clicked"]
        c = _case("nested-%03d" % seed, table, g.plan_kwargs([1]), schema,
                  scan_mode=K.SCAN_NESTED)
        if c:
            out.append(c)
    return out


LSM = dict(uint_cols=["k", "a", "n", "rid"], float_cols=["v"], bool_cols=[],
           key_cols=["k", "s", "n"], first_cols=["a", "s", "v", "rid"],
           lits=[0, 1, 2, 7, 13, 1000, 30000, 65535, 1 << 33])


@suite("lsm")
def lsm_cases():
    """GROUP BY over eventql::PartitionCursor (server/sql/partition_cursor.cc): chains of
    LSM files under the row filters openNextTable builds.  "scan" returns the rid of
    every row the cursor lets through -- the filters themselves, row by row."""
    import lsm_tables
    k, a, n, s, rid = [_c(x) for x in ("k", "a", "n", "s", "rid")]
    named = [
        ("count", dict(select=[Agg("count", Lit(1))], group_by=[])),
        ("group-k", dict(select=[k, Agg("count", Lit(1)), Agg("sum", a)], group_by=[k])),
        # non-aggregates keep the FIRST row in scan order: newest file first
        ("first-row", dict(select=[k, rid, a, s, Agg("count", Lit(1))], group_by=[k])),
        ("string-key", dict(select=[s, Agg("count", Lit(1)), Agg("sum", a)], group_by=[s])),
        ("where", dict(select=[k, Agg("sum", a), Agg("count", Lit(1))], group_by=[k],
                       where=Call("logical_and", Call("gt", a, Lit(30000)),
                                  Call("lt", n, Lit(10_000_000_000))))),
        ("distinct", dict(select=[k, Agg("count_distinct", a)], group_by=[k])),
        ("two-keys", dict(select=[k, s, Agg("count", Lit(1))], group_by=[k, s])),
        ("high-card", dict(select=[Call("mod", rid, Lit(977)), Agg("count", Lit(1)), Agg("sum", a)],
                           group_by=[Call("mod", rid, Lit(977))])),
        # no GROUP BY: the one group's values come from the first row the cursor lets through
        ("global-first-row", dict(select=[rid, s, a, Agg("count", Lit(1))], group_by=[],
                                  where=Call("gt", a, Lit(20000)))),
    ]
    out = []
    for pi, pname in enumerate(sorted(lsm_tables.PARTITIONS)):
        table = "lsm:" + pname
        out.append(dict(id="lsm-%s-scan" % pname, table=table, kw=dict(scan_select=[rid]),
                        sql="select rid from t;", scan_mode=K.SCAN_FLAT))
        for cid, kw in named:
            c = _case("lsm-%s-%s" % (pname, cid), table, kw, lsm_tables.LSM_SCHEMA)
            assert c is not None, cid
            out.append(c)
        for seed in range(4 if pname == "big" else 8):
            g = RefGen(80_000 + 100 * pi + seed, **LSM)
            g.count_cols = ["k", "a", "n", "v"]
            c = _case("lsm-%s-r%02d" % (pname, seed), table, g.plan_kwargs([1]),
                      lsm_tables.LSM_SCHEMA)
            if c:
                out.append(c)
    return out


@suite("siblings")
def sibling_cases():
    """CSTableScan over columns of SIBLING repeated groups of testtbl.cst (cart_items,
    page_view, search_query side by side; result_items below search_query): zipped level by
    level, a group that has run out of slots reads the all-zero SValue"""
    import nested_tables as N
    S = N.SIBLING_SCHEMA
    cq, cp, ci = _c("event.cart_items.quantity"), _c("event.cart_items.price_cents"), _c("event.cart_items.item_id")
    pt, pi = _c("event.page_view.time"), _c("event.page_view.item_id")
    st, sp, sn = _c("event.search_query.time"), _c("event.search_query.page"), _c("event.search_query.num_result_items")
    rp, rc = _c("event.search_query.result_items.position"), _c("event.search_query.result_items.clicked")
    tm, sid = _c("time"), _c("session_id")
    qs = _c("event.search_query.query_string")
    one = Agg("count", Lit(1))
    named = [
        ("two-groups", dict(select=[cq, one, Agg("sum", pt)], group_by=[cq])),
        ("three-groups", dict(select=[sp, one, Agg("sum", cq), Agg("sum", pt)], group_by=[sp])),
        ("with-flat", dict(select=[cq, one, Agg("sum", tm), Agg("sum", pt)], group_by=[cq])),
        ("depth-1-and-2", dict(select=[rp, one, Agg("sum", cq)], group_by=[rp])),
        ("depth-2-and-two-siblings", dict(select=[cq, one, Agg("sum", rp), Agg("sum", pt), Agg("sum", sn)],
                                          group_by=[cq])),
        ("string-keys", dict(select=[ci, one, Agg("sum", pt)], group_by=[ci])),
        ("string-first-rows", dict(select=[cq, pi, ci, sid, one], group_by=[cq])),
        ("two-key-columns", dict(select=[cq, sp, one, Agg("sum", cp)], group_by=[cq, sp])),
        ("bool-leaf", dict(select=[rc, one, Agg("sum", cq), Agg("count_distinct", pt)], group_by=[rc])),
        ("global", dict(select=[one, Agg("sum", cq), Agg("sum", pt), Agg("sum", rp)], group_by=[])),
        # no GROUP BY, non-aggregates: the values of the first emitted row
        # strings of 11 bytes ("kinderstuhl", "sonnenblume"): boxed, they fill the SValue's
        # 16-byte inline buffer, and the PartialGroupBy key hashes the STAG_INLINE bit with them
        ("string-key-of-11-bytes", dict(select=[qs, one, Agg("sum", sn)], group_by=[qs])),
        ("string-key-of-11-bytes-two-keys", dict(select=[qs, sp, one], group_by=[qs, sp])),
        ("global-first-row", dict(select=[cq, ci, sid, one, Agg("sum", pt)], group_by=[])),
        ("global-first-row-one-group", dict(select=[rp, rc, sn, one], group_by=[],
                                            where=Call("gt", rp, Lit(3)))),
        # a real predicate over columns of ONE depth (both groups at depth 1)
        ("where-depth-1", dict(select=[cq, one, Agg("sum", pt)], group_by=[cq],
                               where=Call("logical_or", Call("gt", cq, Lit(1)), Call("gt", pt, Lit(0))))),
    ]
    out = []
    for cid, kw in named:
        c = _case("siblings-" + cid, "testtbl", kw, S, scan_mode=K.SCAN_NESTED)
        assert c is not None, cid
        out.append(c)
    d1 = ["event.cart_items.quantity", "event.cart_items.price_cents", "event.page_view.time",
          "event.search_query.page", "event.search_query.num_result_items", "event.search_query.time"]
    for seed in range(40):
        r = random.Random(90_000 + seed)
        cols = r.sample(d1, r.choice([2, 3, 4]))
        if r.random() < 0.4:
            cols.append("event.search_query.result_items.position")
        if r.random() < 0.4:
            cols.append("time")
        r.shuffle(cols)
        key = _c(cols[0])
        sel = [key, one] + [Agg(r.choice(["sum", "sum", "count", "count_distinct"]), _c(x)) for x in cols[1:]]
        if r.random() < 0.3:
            sel.insert(1, _c(cols[1]))  # a first-row value
        c = _case("siblings-r%02d" % seed, "testtbl", dict(select=sel, group_by=[key]), S,
                  scan_mode=K.SCAN_NESTED)
        if c:
            out.append(c)
    return out


def all_cases():
    return {name: fn() for name, fn in SUITES.items()}


def table_schema(key):
    import nested_tables as N
    if key.startswith("lsm:"):
        import lsm_tables
        return lsm_tables.LSM_SCHEMA
    return {"mixed": T.MIXED_SCHEMA, "ranges": T.RANGES_SCHEMA, "survey": T.SURVEY_SCHEMA,
            "items": N.ITEMS_SCHEMA, "testtbl": N.SIBLING_SCHEMA}[key]
