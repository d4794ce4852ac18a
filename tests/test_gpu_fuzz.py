"""Randomised differential test: random predicates / keys / aggregates over the
mixed-encoding table, HIP path vs oracle.  Seeds are fixed, so failures are
reproducible (`-k "seed3"`)."""
import random

import pytest

import eventql_amd as E
from eventql_amd import capi as K
from eventql_amd.plan import Plan, Col, Lit, Call, If, Agg, CompileError
import oracle_lib as O
import tables as T

pytestmark = pytest.mark.gpu

UINT_COLS = ["k", "a", "b", "n", "p", "k10", "nb", "w"]
FLOAT_COLS = ["v", "nv"]
BOOL_COLS = ["f"]


class Gen:
    def __init__(self, seed):
        self.r = random.Random(seed)

    def uint(self, depth=0):
        r = self.r
        c = r.random()
        if depth > 2 or c < 0.35:
            return Col(r.choice(UINT_COLS))
        if c < 0.5:
            return Lit(r.choice([0, 1, 2, 7, 1000, 30000, 65535, 1 << 40]))
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
            return Col(r.choice(FLOAT_COLS))
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
        if c < 0.55:
            return Col("f")
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
            return Agg("count", Col(r.choice(UINT_COLS + FLOAT_COLS)))
        if c < 0.38:
            return Agg("count_distinct", self.uint())
        if c < 0.6:
            return Agg("sum", self.uint())
        if c < 0.7:
            return Agg("sum", self.flt())
        if c < 0.85:
            return Agg(r.choice(["min", "max"]), r.choice([self.uint(), self.flt()]))
        return Agg("mean", r.choice([self.uint(), self.flt()]))

    def plan_kwargs(self):
        r = self.r
        nkeys = r.choice([0, 1, 1, 1, 2])
        keys = []
        for _ in range(nkeys):
            c = r.random()
            if c < 0.6:
                keys.append(Col(r.choice(["k", "k10", "f", "nb", "n", "s", "ns", "b"])))
            elif c < 0.8:
                keys.append(Call("mod", self.uint(1), Lit(r.choice([3, 17, 1000]))))
            else:
                keys.append(self.boolean(1))
        select = list(keys) + [self.aggregate() for _ in range(r.randint(1, 4))]
        if keys and r.random() < 0.3:
            select.insert(len(keys), Col(r.choice(["a", "v", "s"])))  # first-row value
        kw = dict(select=select, group_by=keys)
        if r.random() < 0.7:
            kw["where"] = self.boolean()
        if r.random() < 0.2:
            kw["row_end"] = r.choice([1, 4097, 131073, 250000])
        kw["groups_hint"] = r.choice([0, 0, 10, 1000, 100000])
        return kw, len(keys) + (1 if len(select) > len(keys) and isinstance(select[len(keys)], Col)
                                and keys else 0)


@pytest.fixture(scope="module")
def mixed(ctx):
    img, _ = T.mixed_table(300_000)
    t = ctx.open_image(img)
    yield t, img
    t.close()


@pytest.mark.parametrize("seed", range(40))
def test_random_plan(mixed, seed):
    t, img = mixed
    g = Gen(seed)
    kw, _ = g.plan_kwargs()
    nkeys = len(kw["group_by"])
    try:
        plan = Plan(T.MIXED_SCHEMA, **kw)
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
