"""Randomised differential tests: random predicates / keys / aggregates, HIP path
vs oracle.  Seeds are fixed, so failures are reproducible (`-k "seed3"`).

Two tables: the mixed-encoding table of the parity tests, and a table whose
bit-packed / plain / LEB128 columns span their full value ranges (values next to
2^17, 2^24, 2^31, 2^32, 2^64), where range-dependent code generation (narrowed
division, shifts, wrap-around) shows."""
import os
import random

import numpy as np
import pytest

import eventql_amd as E
from eventql_amd import capi as K
from eventql_amd.plan import Plan, Col, Lit, Call, If, Agg, CompileError
import oracle_lib as O
import tables as T

pytestmark = pytest.mark.gpu

# EVQL_FUZZ_FROM / EVQL_FUZZ_TO widen the seed range for one-off soak runs
SEEDS = range(int(os.environ.get("EVQL_FUZZ_FROM", "0")), int(os.environ.get("EVQL_FUZZ_TO", "60")))


class Gen:
    two_level_hints = False

    def __init__(self, seed, uint_cols, float_cols, bool_cols, key_cols, first_cols, lits):
        self.seed = seed
        self.r = random.Random(seed)
        self.uint_cols, self.float_cols, self.bool_cols = uint_cols, float_cols, bool_cols
        self.key_cols, self.first_cols, self.lits = key_cols, first_cols, lits

    def uint(self, depth=0):
        r = self.r
        c = r.random()
        if depth > 2 or c < 0.35:
            return Col(r.choice(self.uint_cols))
        if c < 0.5:
            return Lit(r.choice(self.lits))
        if c < 0.9:
            op = r.choice(["add", "sub", "mul", "div", "mod"])
            rhs = self.uint(depth + 1)
            if op in ("div", "mod"):
                rhs = Call("add", rhs, Lit(1))  # never zero... unless it wraps; fine
            return Call(op, self.uint(depth + 1), rhs)
        return If(self.boolean(depth + 1), self.uint(depth + 1), self.uint(depth + 1))

    def flt(self, depth=0):
        r = self.r
        c = r.random()
        if depth > 2 or c < 0.4:
            return Col(r.choice(self.float_cols))
        if c < 0.55:
            return Lit(r.choice([0.0, 1.5, -2.25, 100.0, 8000.5]))
        if c < 0.92:
            return Call(r.choice(["add", "sub", "mul", "div"]), self.flt(depth + 1),
                        self.flt(depth + 1))
        return If(self.boolean(depth + 1), self.flt(depth + 1), self.flt(depth + 1))

    def boolean(self, depth=0):
        r = self.r
        c = r.random()
        if depth > 2 or c < 0.45:
            if r.random() < 0.7:
                return Call(r.choice(["lt", "lte", "gt", "gte", "eq", "neq"]),
                            self.uint(depth + 1), self.uint(depth + 1))
            return Call(r.choice(["lt", "gt", "lte", "gte"]), self.flt(depth + 1),
                        self.flt(depth + 1))
        if c < 0.55 and self.bool_cols:
            return Col(r.choice(self.bool_cols))
        if c < 0.65:
            return Call("neg", self.boolean(depth + 1))
        return Call(r.choice(["logical_and", "logical_or"]), self.boolean(depth + 1),
                    self.boolean(depth + 1))

    def aggregate(self):
        r = self.r
        c = r.random()
        if c < 0.2:
            return Agg("count", Lit(1))
        if c < 0.3:
            return Agg("count", Col(r.choice(self.uint_cols + self.float_cols)))
        if c < 0.38:
            return Agg("count_distinct", self.uint())
        if c < 0.6:
            return Agg("sum", self.uint())
        if c < 0.7:
            return Agg("sum", self.flt())
        if c < 0.85:
            return Agg(r.choice(["min", "max"]), r.choice([self.uint(), self.flt()]))
        return Agg("mean", r.choice([self.uint(), self.flt()]))

    def plan_kwargs(self, row_ends):
        r = self.r
        nkeys = r.choice([0, 1, 1, 1, 2])
        keys = []
        for _ in range(nkeys):
            c = r.random()
            if c < 0.6:
                keys.append(Col(r.choice(self.key_cols)))
            elif c < 0.8:
                keys.append(Call("mod", self.uint(1), Lit(r.choice([3, 13, 17, 1000]))))
            else:
                keys.append(self.boolean(1))
        select = list(keys) + [self.aggregate() for _ in range(r.randint(1, 4))]
        if keys and r.random() < 0.3:
            select.insert(len(keys), Col(r.choice(self.first_cols)))  # first-row value
        kw = dict(select=select, group_by=keys)
        if r.random() < 0.7:
            kw["where"] = self.boolean()
        if r.random() < 0.2:
            kw["row_end"] = r.choice(row_ends)
        kw["groups_hint"] = r.choice([0, 0, 10, 1000, 100000])
        # (100000: one scatter level of the partitioned path; the fuzz tests of this file
        # send every other such plan through coarse + refine instead.  Decided without the
        # generator's random stream: tests/refcases.py derives the committed reference
        # fixtures from this class, seed for seed.)
        if self.two_level_hints and kw["groups_hint"] == 100000 and self.seed % 2:
            kw["groups_hint"] = 3000000
        if self.nrows and r.random() < 0.15:
            # external row filter (LSM skip / update filter, CSTableScan.cc:826-833)
            kw["row_filter"] = np.random.default_rng(r.randrange(1 << 30)).random(self.nrows) < 0.6
        return kw

    nrows = 0  # set by the flat-table tests: enables random row filters


def _integer_only(plan):
    """PartialGroupBy rows are compared as bytes: float states depend on the
    summation order"""
    return all(p.return_type != K.T_FLOAT64 for p in plan.select)


def check_partial(t, img, schema, kw):
    """the same plan as PartialGroupByExpression: (SHA1 key -> saved states /
    encoded values) must equal the oracle's, byte for byte"""
    try:
        plan = Plan(schema, mode=K.MODE_PARTIAL, **kw)
    except CompileError:
        return
    if not _integer_only(plan):
        return
    try:
        exp = O.oracle_run(img, plan)
    except RuntimeError:
        return
    try:
        q = t.query(plan)
    except E.EvqlError as e:
        assert e.code == K.EVQL_ENOTSUP, e
        return
    try:
        got = q.run()
        e = {exp.keys[20 * i:20 * i + 20]: exp.columns[0][i] for i in range(exp.nrows)}
        g = dict(got.rows())
        assert len(g) == got.nrows == exp.nrows
        assert g == e
    finally:
        q.close()


def run_case(t, img, schema, kw, partial_too=False):
    nkeys = len(kw["group_by"])
    if partial_too:
        check_partial(t, img, schema, kw)
    try:
        plan = Plan(schema, **kw)
    except CompileError:
        pytest.skip("type error in the generated expression")
    try:
        exp = O.oracle_run(img, plan)
        exp_err = None
    except RuntimeError as e:
        exp, exp_err = None, str(e)
    try:
        q = t.query(plan)
    except E.EvqlError as e:
        assert e.code == K.EVQL_ENOTSUP, e
        pytest.skip("not lowerable: " + e.msg)
    try:
        if exp_err is not None:
            with pytest.raises(E.EvqlError) as ei:
                q.run()
            assert ("zero" in exp_err) == ("zero" in ei.value.msg)
            return
        got = q.run()
        assert got.nrows == exp.nrows
        # sums of signed float terms may cancel: allow a tiny absolute slack on top
        # of the 1e-6 relative bound
        T.compare_results(got.rows(), exp.rows(), exp.types, key_cols=nkeys, rel=1e-6,
                          abs_tol=1e-3)
    finally:
        q.close()


# ---- the mixed-encoding table -----------------------------------------------------------
MIXED = dict(uint_cols=["k", "a", "b", "n", "p", "k10", "nb", "w"], float_cols=["v", "nv"],
             bool_cols=["f"], key_cols=["k", "k10", "f", "nb", "n", "s", "ns", "b"],
             first_cols=["a", "v", "s"], lits=[0, 1, 2, 7, 1000, 30000, 65535, 1 << 40])


@pytest.fixture(scope="module")
def mixed(ctx):
    img, _ = T.mixed_table(300_000)
    t = ctx.open_image(img)
    yield t, img
    t.close()


@pytest.mark.parametrize("seed", SEEDS)
def test_random_plan(mixed, seed):
    t, img = mixed
    g = Gen(seed, **MIXED)
    g.two_level_hints = True
    g.nrows = 300_000
    kw = g.plan_kwargs([1, 4097, 131073, 250000])
    run_case(t, img, T.MIXED_SCHEMA, kw, partial_too=True)


@pytest.mark.parametrize("seed", range(24))
def test_random_global_group_with_first_row_values(mixed, seed):
    """no GROUP BY, non-aggregate select expressions beside the aggregates: the values of
    the first row that passes WHERE / the row filter (groupby.cc:161-172)"""
    t, img = mixed
    g = Gen(7_000 + seed, **MIXED)
    g.nrows = 300_000
    kw = g.plan_kwargs([1, 4097, 131073, 250000])
    r = random.Random(seed)
    firsts = [Col(c) for c in r.sample(["a", "v", "s", "ns", "n", "nb", "p", "f", "k10"], r.randint(1, 3))]
    kw["select"] = firsts + [e for e in kw["select"] if isinstance(e, Agg)] + [Agg("count", Lit(1))]
    kw["group_by"] = []
    run_case(t, img, T.MIXED_SCHEMA, kw, partial_too=True)


# ---- nested (Dremel) scans ------------------------------------------------------------------
def _nested_gens():
    import nested_tables as N
    items = dict(uint_cols=["id", "items.position", "items.price"], float_cols=["score"],
                 bool_cols=[], key_cols=["items.position", "id"], first_cols=["items.price", "id"],
                 lits=[0, 1, 3, 7, 1000, 50000, 1 << 33])
    fixture = dict(uint_cols=["time", "event.search_query.time",
                              "event.search_query.num_result_items",
                              "event.search_query.result_items.position"],
                   float_cols=[], bool_cols=["event.search_query.result_items.clicked"],
                   key_cols=["event.search_query.result_items.position",
                             "event.search_query.num_result_items",
                             "event.search_query.result_items.clicked"],
                   first_cols=["time", "event.search_query.num_result_items"],
                   lits=[0, 1, 2, 6, 10, 1438055327])
    return N, items, fixture


@pytest.fixture(scope="module")
def nested(ctx):
    N, _, _ = _nested_gens()
    img_items, _ = N.items_table(50_000)
    img_fix = N.testtbl_v2()
    ti, tf = ctx.open_image(img_items), ctx.open_image(img_fix)
    yield (ti, img_items), (tf, img_fix)
    ti.close()
    tf.close()


class NestedGen(Gen):
    """float productions without a float column use literals; WHERE draws from the
    leaf-level columns only (a WHERE over columns of different repetition depth
    is answered with ENOTSUP, see planner.cc)"""
    leaf_uint, leaf_bool = (), ()

    def flt(self, depth=0):
        if not self.float_cols:
            return Lit(self.r.choice([0.0, 1.5, -2.25, 100.0]))
        return Gen.flt(self, depth)

    def plan_kwargs(self, row_ends):
        # a WHERE is lowered only when every scan column has the same repetition
        # depth (the reference's reset quirk): half of the plans use leaf columns
        # only and keep their WHERE, the others use all columns without one
        if self.r.random() < 0.5:
            self.uint_cols = list(self.leaf_uint)
            self.float_cols, self.bool_cols = [], list(self.leaf_bool)
            self.key_cols = list(self.leaf_uint) + list(self.leaf_bool)
            self.first_cols = list(self.leaf_uint)
            return Gen.plan_kwargs(self, row_ends)
        kw = Gen.plan_kwargs(self, row_ends)
        kw.pop("where", None)
        return kw


@pytest.mark.parametrize("seed", SEEDS)
def test_random_nested_plan(nested, seed):
    N, items, fixture = _nested_gens()
    (ti, img_items), (tf, img_fix) = nested
    which = seed % 2
    t, img, schema, cols = ((ti, img_items, N.ITEMS_SCHEMA, items) if which == 0 else
                            (tf, img_fix, N.NESTED_SCHEMA, fixture))
    g = NestedGen(2000 + seed, **cols)
    g.two_level_hints = True
    if which == 0:
        g.leaf_uint, g.leaf_bool = ["items.position", "items.price"], []
    else:
        g.leaf_uint = ["event.search_query.result_items.position"]
        g.leaf_bool = ["event.search_query.result_items.clicked"]
    kw = g.plan_kwargs([1])
    kw.pop("row_end", None)  # row ranges do not apply to nested scans
    kw["scan_mode"] = K.SCAN_NESTED
    run_case(t, img, schema, kw)


@pytest.mark.parametrize("seed", SEEDS)
def test_random_within_record_plan(nested, seed):
    """AGGREGATE_WITHIN_RECORD_FLAT: a random list of per-record count / sum over
    columns of every depth (or literals), random operators above it over `$i`"""
    import random
    from eventql_amd.plan import count, sum_, col, lit
    N, items, fixture = _nested_gens()
    (ti, img_items), (tf, img_fix) = nested
    which = seed % 2
    t, img, schema, cols = ((ti, img_items, N.ITEMS_SCHEMA, items) if which == 0 else
                            (tf, img_fix, N.NESTED_SCHEMA, fixture))
    r = random.Random(7000 + seed)
    inner = []
    for _ in range(r.randint(1, 5)):
        c = col(r.choice(cols["uint_cols"]))
        inner.append(r.choice([lambda: count(c), lambda: sum_(c), lambda: count(1),
                               lambda: sum_(lit(r.choice([1, 3, 1000]))),
                               lambda: count(col(r.choice(cols["uint_cols"] + cols["bool_cols"])))
                               ])())
    # (a record scan needs at least one column)
    inner.append(count(col(r.choice(cols["uint_cols"]))))
    outs = ["$%d" % i for i in range(len(inner))]
    g = Gen(8000 + seed, uint_cols=outs, float_cols=[], bool_cols=[], key_cols=outs,
            first_cols=outs, lits=[0, 1, 2, 5, 9, 40, 1000])
    g.flt = lambda depth=0: Lit(g.r.choice([0.0, 1.5, -2.25, 100.0]))
    kw = g.plan_kwargs([1])
    kw.pop("row_end", None)
    kw.pop("where", None)
    kw["scan_select"] = inner
    kw["scan_mode"] = K.SCAN_NESTED_WITHIN_RECORD
    run_case(t, img, schema, kw)


# ---- full-range columns -------------------------------------------------------------------
RANGES_SCHEMA, RANGES, ranges_table = T.RANGES_SCHEMA, T.RANGES, T.ranges_table


@pytest.fixture(scope="module")
def ranges(ctx):
    img = ranges_table()
    t = ctx.open_image(img)
    yield t, img
    t.close()


@pytest.mark.parametrize("seed", SEEDS)
def test_random_plan_full_range_columns(ranges, seed):
    t, img = ranges
    g = Gen(1000 + seed, **RANGES)
    g.two_level_hints = True
    g.nrows = 200_000
    kw = g.plan_kwargs([1, 4097, 131073, 150000])
    run_case(t, img, RANGES_SCHEMA, kw, partial_too=True)
