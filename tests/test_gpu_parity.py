"""Parity of the HIP path (through the C ABI) against the oracle on identical
inputs.  Integer / string / NULL results must be bit-exact; float64 sums and
means within 1e-6 relative (BASELINE.json north_star)."""
import json
import os

import numpy as np
import pytest

import eventql_amd as E
from eventql_amd import capi as K, synth, bench_plans as B
from eventql_amd.plan import Plan, col, count, sum_, min_, max_, mean, If, lit, Call, Agg
import oracle_lib as O
import tables as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mixed(ctx):
    img, c = T.mixed_table(300_000)
    t = ctx.open_image(img)
    yield t, img, c
    t.close()


def check(t, img, key_cols=1, schema=None, **kw):
    plan = Plan(schema or T.MIXED_SCHEMA, **kw)
    exp = O.oracle_run(img, plan)
    q = t.query(plan)
    try:
        got = q.run()
        assert [q.column_type(i) for i in range(q.column_count())] == exp.types
        assert got.nrows == exp.nrows, (got.nrows, exp.nrows)
        T.compare_results(got.rows(), exp.rows(), exp.types, key_cols=key_cols)
        st = q.stats()
        assert st["rows_passed"] == exp.rows_passed
        return got, exp, st
    finally:
        q.close()


W = (col("a") > 30000) & (col("b") < 30000)


@pytest.mark.parametrize("key", ["k", "k10", "p", "b", "t", "f"])
def test_group_by_every_direct_and_decoded_encoding(mixed, key):
    """keys from LEB128 (decoded to SoA), narrow bit-packed, UINT32_PLAIN,
    UINT64_PLAIN, DATETIME/LEB128 and BOOLEAN columns"""
    t, img, _ = mixed
    check(t, img, select=[col(key), sum_(col("a")), count(1), sum_(col("v")), sum_(col("p"))],
          group_by=[col(key)], where=W)


def test_config_shapes(mixed):
    t, img, _ = mixed
    check(t, img, select=[col("k"), sum_(col("v")), count(1)], group_by=[col("k")],
          groups_hint=1000)
    check(t, img, select=[col("k"), sum_(col("v")), count(1), sum_(col("b"))],
          group_by=[col("k")], where=W, groups_hint=1000)


@pytest.mark.parametrize("hint", [0, 2, 4, 10, 1000, 5000, 100000])
def test_group_table_variants(mixed, hint):
    """LDS table sizes, the one-workgroup-per-CU variant and the HBM-only table; hints of
    2 .. 4 select the four-entry lane-private accumulator cache (here with 1000 groups:
    every row evicts)"""
    t, img, _ = mixed
    _, _, st = check(t, img, select=[col("k"), sum_(col("a")), count(1), min_(col("b")),
                                     max_(col("v"))],
                     group_by=[col("k")], where=col("a") > 1000, groups_hint=hint)
    assert st["num_groups"] == 1000


@pytest.mark.parametrize("ngroups", [1, 2, 3, 4])
def test_very_few_groups(ctx, ngroups):
    """1 .. 4 groups, first-row values, min / max / float sums, nullable keys: the rows of
    a lane are combined in its accumulator cache and reach the LDS table only at the end"""
    n = 700_001
    i = np.arange(n, dtype=np.uint64)
    w = E.Writer([dict(name="g", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN,
                       dlevel_max=1),
                  dict(name="a", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
                  dict(name="v", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754)])
    g = (i * np.uint64(2654435761) >> np.uint64(7)) % np.uint64(ngroups)
    w.put("g", g, present=(i % 11 != 3).astype(np.uint8))
    w.put("a", (i * np.uint64(40503)) % np.uint64(65521))
    w.put("v", ((i * np.uint64(7919)) % np.uint64(4096)).astype(np.float64) / 8.0)
    w.commit(n)
    img = w.image()
    w.close()
    t = ctx.open_image(img)
    S = dict(g=K.T_UINT64, a=K.T_UINT64, v=K.T_FLOAT64)
    try:
        for hint in (0, ngroups + 1):
            check(t, img, select=[col("g"), col("a"), count(1), sum_(col("a")), min_(col("v")),
                                  max_(col("a")), sum_(col("v"))],
                  group_by=[col("g")], where=col("a") > 100, groups_hint=hint, schema=S)
    finally:
        t.close()


def test_high_cardinality(mixed):
    t, img, c = mixed
    # 300k distinct 64-bit keys; LDS table overflows into the HBM table
    _, _, st = check(t, img, select=[col("w"), count(1), sum_(col("a"))], group_by=[col("w")])
    assert st["num_groups"] == len(set(c["w"].tolist()))
    check(t, img, select=[col("w"), count(1), sum_(col("a"))], group_by=[col("w")],
          groups_hint=400000)
    check(t, img, select=[col("b"), count(1), sum_(col("v")), max_(col("a"))],
          group_by=[col("b")], groups_hint=70000)


def test_partitioned_high_cardinality_path(mixed):
    """radix-partition + per-bucket LDS aggregation (count / scatter / aggregate
    kernels): a large groups_hint selects it whatever the data holds"""
    t, img, c = mixed
    for kw in (
        dict(select=[col("w"), count(1), sum_(col("a")), sum_(col("v"))], group_by=[col("w")]),
        dict(select=[col("k"), count(1), sum_(col("a")), min_(col("v")), max_(col("b")),
                     mean(col("n")), min_(col("nv"))], group_by=[col("k")], where=W),
        # first-row semantics and hashed (multi-column / string) identities
        dict(select=[col("k"), col("a"), col("s"), count(1)], group_by=[col("k")]),
        dict(select=[col("k"), col("f"), count(1), sum_(col("p"))], group_by=[col("k"), col("f")],
             key_cols=2),
        dict(select=[col("s"), count(1), max_(col("a"))], group_by=[col("s")]),
        dict(select=[col("b"), count(1), sum_(col("v"))], group_by=[col("b")], row_end=123_457),
    ):
        kc = kw.pop("key_cols", 1)
        # 256 buckets: one scatter level; 1024 / 4096 buckets: coarse + refine
        for hint in (100_000, 300_000, 3_000_000):
            got, exp, st = check(t, img, key_cols=kc, groups_hint=hint, **kw)
    # the kernels really are the partitioned ones
    plan = Plan(T.MIXED_SCHEMA, select=[col("w"), count(1)], group_by=[col("w")],
                groups_hint=300_000)
    q = t.query(plan)
    assert "evql_part_scatter" in q.kernel_source()
    q.close()
    q = t.query(Plan(T.MIXED_SCHEMA, select=[col("w"), count(1)], group_by=[col("w")],
                     groups_hint=3_000_000))
    assert "evql_part_refine" in q.kernel_source()
    q.close()
    q = t.query(Plan(T.MIXED_SCHEMA, select=[col("w"), count(1)], group_by=[col("w")],
                     groups_hint=100_000))
    assert "evql_part_refine" not in q.kernel_source()
    q.close()
    q = t.query(Plan(T.MIXED_SCHEMA, select=[col("w"), count(1)], group_by=[col("w")],
                     groups_hint=1000))
    assert "evql_part_scatter" not in q.kernel_source()
    q.close()


def test_partition_tuple_widths(mixed):
    """tuple members that the table's column maxima bound below 2^32 travel as 4 bytes
    through scatter / refine / aggregate (runtime.cc choose_tuple_widths): the layouts
    the generator picks, and every narrow member kind against the oracle"""
    import re
    t, img, c = mixed

    def u32_words(**kw):
        q = t.query(Plan(T.MIXED_SCHEMA, groups_hint=3_000_000, **kw))
        try:
            return int(re.search(r"#define EVQL_TUPLE_U32 (\d+)", q.kernel_source()).group(1))
        finally:
            q.close()

    a, b, k, w, v = col("a"), col("b"), col("k"), col("w"), col("v")
    # key < 2^32, a < 2^32, float sum: 1 + 1 + 2 words (one 16-byte access per tuple)
    assert u32_words(select=[b, sum_(a), count(1), sum_(v)], group_by=[b]) == 4
    # 64-bit key values stay 8 bytes
    assert u32_words(select=[w, sum_(a), count(1), sum_(v)], group_by=[w]) == 6
    # a product that may pass 2^32 stays wide; a - b may wrap: wide
    assert u32_words(select=[b, sum_(a * b * 4)], group_by=[b]) == 4   # 1 + 2, padded
    assert u32_words(select=[b, sum_(a * 3)], group_by=[b]) == 2
    assert u32_words(select=[b, sum_(a - b)], group_by=[b]) == 4
    # hashed identity (2 x 2) + row (1) + narrow sum (1)
    assert u32_words(select=[k, b, sum_(a)], group_by=[k, b]) == 6
    for kw in (
        dict(select=[b, sum_(a), min_(a), max_(a), min_(col("nb")), mean(a), count(1)], group_by=[b]),
        dict(select=[b, sum_(a * 3 + 1), max_(a % 7), sum_(a * b), sum_(v)], group_by=[b], where=W),
        dict(select=[a % 1000, min_(b), sum_(col("p")), max_(col("p"))], group_by=[a % 1000]),
        dict(select=[b, col("s"), col("w"), max_(a)], group_by=[b]),
        dict(select=[col("f"), b, min_(a), sum_(col("n"))], group_by=[col("f"), b], key_cols=2),
    ):
        kc = kw.pop("key_cols", 1)
        for hint in (100_000, 3_000_000):
            check(t, img, key_cols=kc, groups_hint=hint, **kw)
    # exact float sums: the low 31 bits travel as 4 bytes
    kw = dict(select=[b, sum_(v), sum_(a), count(1)], group_by=[b])
    exp = O.oracle_run(img, Plan(T.MIXED_SCHEMA, **kw))
    q = t.query(Plan(T.MIXED_SCHEMA, float_sum_mode=K.FLOAT_SUM_EXACT, groups_hint=3_000_000, **kw))
    assert "evql_part_refine" in q.kernel_source()
    T.compare_results(q.run().rows(), exp.rows(), exp.types, rel=1e-12)
    q.close()


def test_wide_plans(mixed):
    """many aggregates: 20+ state words per slot leave the LDS table only a few hundred
    slots, so 1000 groups already take the partitioned path (wide tuples: 2 per thread
    in aggregate, 1 per thread and chunk in refine), and the tile loop runs with fewer
    unroll steps than the default (register budget)"""
    import re
    t, img, c = mixed
    a, b, k, v, p, n = col("a"), col("b"), col("k"), col("v"), col("p"), col("n")
    sel = [k, sum_(a), sum_(b), sum_(a * b), min_(a), max_(b), mean(v), sum_(v), count(1),
           sum_(If(a > b, a - b, b - a)), max_(v * 2.0), min_(v), mean(a), mean(b), sum_(a % 7),
           sum_(b % 13), min_(col("nb")), max_(p), sum_(n)]
    for hint in (0, 1000, 3_000_000):
        got, exp, st = check(t, img, select=sel, group_by=[k], where=(a > 100) & (b < 65000),
                             groups_hint=hint)
        assert st["num_groups"] == 1000
    q = t.query(Plan(T.MIXED_SCHEMA, select=sel, group_by=[k], groups_hint=1000))
    src = q.kernel_source()
    q.close()
    assert "evql_part_scatter" in src            # 1000 groups > the LDS slots of this plan
    assert int(re.search(r"#define EVQL_UNROLL (\d+)", src).group(1)) < 4
    assert int(re.search(r"#define EVQL_TUPLE_U32 (\d+)", src).group(1)) > 20
    # the same select list over a two-column key with first-row values
    check(t, img, key_cols=2, select=[k, col("f"), col("s")] + sel[1:], group_by=[k, col("f")],
          groups_hint=5000)


def test_global_aggregates(mixed):
    t, img, _ = mixed
    check(t, img, key_cols=0, select=[count(1)])
    check(t, img, key_cols=0, select=[count(1), sum_(col("a")), sum_(col("v")), min_(col("a")),
                                      max_(col("v")), mean(col("b")), mean(col("v"))], where=W)
    # zero passing rows => zero result rows (groupby.cc:183,192)
    got, _, _ = check(t, img, key_cols=0, select=[count(1), sum_(col("a"))],
                      where=col("v") > 8000000.5)
    assert got.nrows == 0


def test_nullable_columns(mixed):
    t, img, _ = mixed
    # NULL contributes 0 to sum, is counted by count(x), skipped by min/max/mean
    check(t, img, select=[col("k"), sum_(col("n")), count(col("n")), min_(col("n")),
                          max_(col("n")), mean(col("n")), sum_(col("nv")), min_(col("nv")),
                          mean(col("nv")), sum_(col("nb")), max_(col("nb"))],
          group_by=[col("k")])
    # NULL compares as 0 in predicates
    check(t, img, key_cols=0, select=[count(1)], where=col("n") > 5)
    check(t, img, key_cols=0, select=[count(1)], where=col("nv") < 1.0)
    # NULL keys form their own group, distinct from 0
    got, _, _ = check(t, img, select=[col("nb"), count(1), sum_(col("a"))], group_by=[col("nb")])
    assert None in [r[0] for r in got.rows()] and 0 in [r[0] for r in got.rows()]
    check(t, img, select=[col("nv"), count(1)], group_by=[col("nv")], groups_hint=20000)
    # only-NULL input => min/mean are NULL
    check(t, img, key_cols=0, select=[min_(col("n")), mean(col("n")), count(1)],
          where=col("n").eq(0))


def test_string_keys(mixed):
    t, img, _ = mixed
    check(t, img, select=[col("s"), count(1), sum_(col("a"))], group_by=[col("s")])
    got, _, _ = check(t, img, select=[col("ns"), count(1), sum_(col("v"))],
                      group_by=[col("ns")])
    assert None in [r[0] for r in got.rows()]
    check(t, img, key_cols=2, select=[col("k"), col("s"), count(1)],
          group_by=[col("k"), col("s")])


def test_string_predicates(mixed):
    """boolean.cc string compares (eq/neq = memcmp, ordering = strncmp then
    length; tags ignored, NULL = ""), bytewise in the kernel over the
    STRING_PLAIN page stream"""
    t, img, _ = mixed
    s, ns, a, k = col("s"), col("ns"), col("a"), col("k")
    check(t, img, key_cols=0, select=[count(1), sum_(a)], where=s.eq("g5"))
    check(t, img, key_cols=0, select=[count(1), sum_(a)], where=s.neq("g77") & (a > 30000))
    check(t, img, select=[k, count(1)], group_by=[k], where=(s < "g5") | (s >= "g95"))
    check(t, img, key_cols=0, select=[count(1)], where=ns.eq(""))          # NULL strings read ""
    check(t, img, key_cols=0, select=[count(1)], where=ns > "s3")
    check(t, img, key_cols=0, select=[count(1)], where=ns <= "s30")
    check(t, img, key_cols=0, select=[count(1)], where=s < ns)             # column vs column
    check(t, img, key_cols=0, select=[count(1)], where=Call("cmp", s, ns) > 0)
    check(t, img, select=[s < "g5", count(1), sum_(If(ns.eq("s7"), a, 0))], group_by=[s < "g5"])
    check(t, img, select=[ns, count(1)], group_by=[ns], where=ns.neq("s1") & s.neq("g1"))
    check(t, img, key_cols=0, select=[count(1)], where=s.eq("no such value"))


def test_string_predicates_over_page_straddling_values(ctx):
    """values longer than a few bytes, embedded NULs, empty strings, and 2 MiB of
    string data so that values straddle the 512 KiB pages"""
    n = 60_000
    rng = np.random.default_rng(5)
    words = [b"", b"a", b"a\0b", b"a\0c", b"ab", b"abc" * 9, b"\xff\xfe", b"zebra" * 13, b"a\0"]
    vals = [words[i] for i in rng.integers(0, len(words), n)]
    w = E.Writer([dict(name="s", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN),
                  dict(name="x", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN)])
    w.put("s", vals)
    w.put("x", np.arange(n, dtype=np.uint64))
    w.commit(n)
    img = w.image()
    w.close()
    assert len(img) > 3 * 512 * 1024
    t = ctx.open_image(img)
    S = dict(s=K.T_STRING, x=K.T_UINT64)
    s = col("s")
    for lit_ in (b"", b"a", b"a\0b", b"a\0c", b"ab", b"abc" * 9, b"abd", b"\xff", b"zebra" * 13):
        for op in ("eq", "neq", "lt", "lte", "gt", "gte"):
            plan = Plan(S, select=[count(1), sum_(col("x"))], where=Call(op, s, lit(lit_)))
            exp = O.oracle_run(img, plan)
            q = t.query(plan)
            assert q.run().rows() == exp.rows(), (lit_, op)
            q.close()
    t.close()


def test_multi_column_keys_and_first_row(mixed):
    t, img, _ = mixed
    check(t, img, key_cols=2, select=[col("k"), col("f"), count(1), sum_(col("a"))],
          group_by=[col("k"), col("f")])
    check(t, img, key_cols=2, select=[col("k"), col("nb"), count(1)],
          group_by=[col("k"), col("nb")], groups_hint=200000)
    # non-aggregate select expressions: value of the group's FIRST row in scan order
    check(t, img, select=[col("k"), col("a"), col("v"), col("s"), count(1)],
          group_by=[col("k")])
    check(t, img, select=[col("k"), col("a") + col("b"), count(1)], group_by=[col("k")],
          where=W)
    # group by an expression
    check(t, img, select=[col("a") % 10, count(1), sum_(col("b"))], group_by=[col("a") % 10])


def test_global_group_reads_its_first_passing_row(mixed):
    """no GROUP BY, non-aggregate select expressions: the one group's values are those of
    the first row that passes WHERE (groupby.cc:161-172) -- found by the round-3 soak: the
    ungrouped kernel path never recorded that row and the gather read outside the table"""
    t, img, _ = mixed
    a, k, v = col("a"), col("k"), col("v")
    for where in (None, (k < 60) & (a > 1000), a > 65000, (a % 9973).eq(17), v > 9.0e9):
        kw = dict(where=where) if where is not None else {}
        check(t, img, key_cols=0, select=[col("ns"), col("s"), col("p"), col("nv"), count(1), sum_(a)], **kw)
        check(t, img, key_cols=0, select=[Call("concat", col("ns"), "x"), Call("to_string", col("nb")),
                                          count(1)], **kw)
    # a row range and a row filter move the first row
    check(t, img, key_cols=0, select=[col("s"), col("a"), count(1)], where=a > 60000, row_end=250000)
    flt = np.zeros(300_000, dtype=bool)
    flt[123_457::3] = True
    check(t, img, key_cols=0, select=[col("s"), col("n"), count(1)], where=a > 50000, row_filter=flt)
    # the PartialGroupBy row of the global group carries the encoded first-row values
    plan = Plan(T.MIXED_SCHEMA, mode=K.MODE_PARTIAL, select=[col("ns"), col("p"), count(1), sum_(a)],
                where=(k < 60) & (a > 1000))
    exp = O.oracle_run(img, plan)
    q = t.query(plan)
    got = q.run()
    assert got.nrows == exp.nrows == 1
    assert dict(got.rows()) == {exp.keys[:20]: exp.columns[0][0]}
    q.close()


def test_expressions(mixed):
    t, img, _ = mixed
    a, b, v, k, p = col("a"), col("b"), col("v"), col("k"), col("p")
    check(t, img, select=[k, sum_(If(a > b, a - b, b - a)), sum_(a * b + 7), sum_(v * 1.5 - 2.0),
                          sum_(v / (v + 1.0)), max_(a / (b + 1)), sum_(p % 13), count(1)],
          group_by=[k], where=((a + b) % 7 > 2) | ~(v >= 100.0))
    # uint64 wrap-around and UINT64 -> INT64 conversion with signed compare
    check(t, img, select=[k, sum_(a - 40000), sum_(b * 281474976710656), count(1)], group_by=[k])
    check(t, img, select=[k, sum_(Call("to_int64", a) - 40000), min_(Call("to_int64", a) - 40000),
                          count(1)],
          group_by=[k], where=(a - lit(-5)) > lit(-1))
    # post-aggregate arithmetic; the single-instance quirk (sum(a)+sum(b) = 2*sum(a))
    check(t, img, select=[k, sum_(a) + 1, sum_(a) + sum_(b), sum_(v) * 2.0], group_by=[k])
    # (`sum(a) + k` is not a valid reference plan: GroupByExpression::nextBatch runs
    # method_call with argc = 0, so an X_INPUT there trips vm.cc:135's assert)
    # cmp / eq / neq on floats and bools
    check(t, img, select=[col("f"), count(1), sum_(If(col("f"), 1, 0))], group_by=[col("f")],
          where=v.neq(0.0) & Call("cmp", a, b).eq(lit(-1)))
    # float division by zero is permitted (math.cc:166-170), pow / mod via libm
    check(t, img, key_cols=0, select=[count(1), max_(Call("pow", v, 0.5)), sum_(Call("mod", v, 3.0))],
          where=(v / (v - v)) > 1.0)


def test_division_by_zero_is_reported(mixed):
    t, img, _ = mixed
    plan = Plan(T.MIXED_SCHEMA, select=[count(1)], where=(col("a") / (col("b") - col("b"))) > 1)
    with pytest.raises(RuntimeError, match="division by zero"):
        O.oracle_run(img, plan)
    q = t.query(plan)
    with pytest.raises(E.EvqlError) as ei:
        q.run()
    assert ei.value.code == K.EVQL_ERUNTIME and "division by zero" in ei.value.msg
    q.close()
    # ... but only for rows that are actually evaluated (IF guards the division)
    check(t, img, select=[col("k"), sum_(If(col("b") > 0, col("a") / col("b"), 0))],
          group_by=[col("k")])


def test_row_filter_and_row_range(mixed):
    t, img, _ = mixed
    n = 300_000
    rng = np.random.default_rng(7)
    keep = rng.random(n) < 0.3
    check(t, img, select=[col("k"), count(1), sum_(col("a"))], group_by=[col("k")],
          row_filter=keep.astype(np.uint8), where=col("b") > 1000)
    # a filter shorter than the table drops the rows beyond it
    check(t, img, key_cols=0, select=[count(1)], row_filter=np.ones(1000, np.uint8))
    # prefix scans end on arbitrary rows
    for end in (1, 2047, 2048, 2049, 65535, 65537, 131073, 299_999):
        check(t, img, select=[col("k"), count(1), sum_(col("b")), sum_(col("p"))],
              group_by=[col("k")], row_end=end)


def test_row_ranges_partition_the_table(mixed):
    """partition slices: partial aggregates over [0,m) and [m,n) merged on the
    device equal the whole-table result (PartialGroupBy -> GroupByMerge)"""
    import torch
    t, img, _ = mixed
    n, m = 300_000, 123_457
    kw = dict(select=[col("k"), count(1), sum_(col("a")), min_(col("b")), max_(col("v")),
                      sum_(col("v"))], group_by=[col("k")], where=col("a") > 5000)
    whole = Plan(T.MIXED_SCHEMA, **kw)
    exp = O.oracle_run(img, whole)
    qa = t.query(Plan(T.MIXED_SCHEMA, row_end=m, **kw))
    qb = t.query(Plan(T.MIXED_SCHEMA, row_begin=m, **kw))
    qa.execute()
    qb.execute()
    rw = qb.record_words()
    buf = torch.zeros(4096 * rw, dtype=torch.int64, device="cuda")
    cnt = qb.export_groups(buf.data_ptr(), 4096)
    assert cnt == 1000
    qa.import_groups(buf.data_ptr(), cnt)
    got = qa.fetch_all()
    T.compare_results(got.rows(), exp.rows(), exp.types)
    qa.close()
    qb.close()


def test_import_grows_the_group_table(mixed):
    """a merge target never fills up half way (GroupByMergeExpression's map just grows,
    groupby.cc:528-637): the table is rebuilt with room for the incoming groups first --
    here 8 slices of ~37,000 distinct 64-bit keys each into a table that started with
    65,536 slots, and into an empty `reset` target"""
    import torch
    t, img, _ = mixed
    n, parts = 300_000, 8
    kw = dict(select=[col("w"), count(1), sum_(col("a")), min_(col("b"))], group_by=[col("w")])
    exp = O.oracle_run(img, Plan(T.MIXED_SCHEMA, **kw))
    cut = [n * i // parts for i in range(parts + 1)]
    qs = [t.query(Plan(T.MIXED_SCHEMA, row_begin=cut[i], row_end=cut[i + 1], groups_hint=1000, **kw))
          for i in range(parts)]
    for q in qs:
        q.execute()
    rw = qs[0].record_words()
    buf = torch.zeros(60_000 * rw, dtype=torch.int64, device="cuda")
    target = t.query(Plan(T.MIXED_SCHEMA, groups_hint=1000, **kw))
    target.reset()
    for dst in (qs[0], target):
        for q in qs[1:]:
            cnt = q.export_groups(buf.data_ptr(), 60_000)
            dst.import_groups(buf.data_ptr(), cnt)
    # qs[0] now holds everything; target holds slices 1..7 -- add slice 0 through a fresh
    # scan of that range
    q0 = t.query(Plan(T.MIXED_SCHEMA, row_begin=cut[0], row_end=cut[1], groups_hint=1000, **kw))
    q0.execute()
    cnt = q0.export_groups(buf.data_ptr(), 60_000)
    target.import_groups(buf.data_ptr(), cnt)
    for dst in (qs[0], target):
        got = dst.fetch_all(1 << 20)
        assert got.nrows == exp.nrows
        T.compare_results(got.rows(), exp.rows(), exp.types)
    for q in qs + [target, q0]:
        q.close()


def test_count_distinct_through_export_and_import(mixed):
    """callers that move partial aggregates themselves (evql_query_export / import_groups):
    count_distinct travels as its (group, value) pairs (evql_query_export_pairs /
    _import_pairs) and is counted again in the target's set -- 6 row slices merged into the
    first one and into an empty `reset` target, against the oracle on the whole table"""
    import torch
    from eventql_amd.plan import count_distinct as cd
    t, img, _ = mixed
    n, parts = 300_000, 6
    kw = dict(select=[col("k"), cd(col("a") % 977), count(1), cd(col("b")), sum_(col("a"))],
              group_by=[col("k")])
    exp = O.oracle_run(img, Plan(T.MIXED_SCHEMA, **kw))
    cut = [n * i // parts for i in range(parts + 1)]
    qs = [t.query(Plan(T.MIXED_SCHEMA, row_begin=cut[i], row_end=cut[i + 1], groups_hint=1000, **kw))
          for i in range(parts)]
    for q in qs:
        q.execute()
    assert lib_distinct(qs[0]) == 2
    rw = qs[0].record_words()
    buf = torch.zeros(4096 * rw, dtype=torch.int64, device="cuda")
    target = t.query(Plan(T.MIXED_SCHEMA, groups_hint=1000, **kw))
    target.reset()
    for q in qs[1:]:
        cnt = q.export_groups(buf.data_ptr(), 4096)
        qs[0].import_groups(buf.data_ptr(), cnt)
        for which in range(2):
            npairs = q.export_pairs(which, None, 0)
            pbuf = torch.zeros(max(1, npairs) * 3, dtype=torch.int64, device="cuda")
            assert q.export_pairs(which, pbuf.data_ptr(), npairs) == npairs
            qs[0].import_pairs(which, pbuf.data_ptr(), npairs)
    fresh = [t.query(Plan(T.MIXED_SCHEMA, row_begin=cut[i], row_end=cut[i + 1], groups_hint=1000, **kw))
             for i in range(parts)]
    for q in fresh:
        q.execute()
        cnt = q.export_groups(buf.data_ptr(), 4096)
        target.import_groups(buf.data_ptr(), cnt)
        for which in range(2):
            npairs = q.export_pairs(which, None, 0)
            pbuf = torch.zeros(max(1, npairs) * 3, dtype=torch.int64, device="cuda")
            q.export_pairs(which, pbuf.data_ptr(), npairs)
            target.import_pairs(which, pbuf.data_ptr(), npairs)
    for dst in (qs[0], target):
        got = dst.fetch_all(1 << 20)
        assert got.nrows == exp.nrows
        T.compare_results(got.rows(), exp.rows(), exp.types)
    for q in qs + fresh + [target]:
        q.close()


def lib_distinct(q):
    return E.lib().evql_query_distinct_aggregates(q.h)


def test_partial_group_by_wire_rows(mixed):
    """EVQL_MODE_PARTIAL = PartialGroupByExpression::nextBatch (groupby.cc:438-472):
    (SHA1 group key, concatenated saved states / encoded SValues), byte for byte"""
    t, img, _ = mixed
    for kw in (dict(select=[col("k"), sum_(col("a")), count(1)], group_by=[col("k")],
                    where=col("a") >= 0),
               dict(select=[col("k"), col("s"), count(1), sum_(col("b"))],
                    group_by=[col("k"), col("s")]),
               dict(select=[col("nb"), count(1), max_(col("a")), mean(col("a"))],
                    group_by=[col("nb")]),
               dict(select=[count(1), sum_(col("a"))])):
        plan = Plan(T.MIXED_SCHEMA, mode=K.MODE_PARTIAL, **kw)
        exp = O.oracle_run(img, plan)
        q = t.query(plan)
        got = q.run()
        assert q.column_count() == 2 and got.types == [K.T_STRING, K.T_STRING]
        e = {exp.keys[20 * i:20 * i + 20]: exp.columns[0][i] for i in range(exp.nrows)}
        g = dict(got.rows())
        assert len(g) == got.nrows == exp.nrows
        assert g == e
        q.close()


def test_partial_rows_from_the_gpu_merge_to_the_final_result(mixed):
    """PartialGroupBy on the device over three row ranges -> wire rows ->
    GroupByMergeExpression (evql_merge_*, groupby.cc:528-672) == the oracle's
    whole-table GROUP BY; also merged by the oracle's own merge restatement"""
    t, img, _ = mixed
    cuts = [(0, 100_001), (100_001, 222_222), (222_222, 300_000)]
    for kw, nkeys in (
            (dict(select=[col("k"), sum_(col("a")), count(1), min_(col("nb")), mean(col("v"))],
                  group_by=[col("k")], where=W), 1),
            (dict(select=[col("ns"), col("f"), count(1), sum_(col("v")), max_(col("w"))],
                  group_by=[col("ns"), col("f")]), 2)):
        whole = Plan(T.MIXED_SCHEMA, **kw)
        exp = O.oracle_run(img, whole)
        m = E.Merge(whole)
        frames = []
        for lo, hi in cuts:
            q = t.query(Plan(T.MIXED_SCHEMA, mode=K.MODE_PARTIAL, row_begin=lo, row_end=hi, **kw))
            q.execute()
            keys, datas = [], []
            while True:
                n, raw = q.next_batch(1024)
                if n == 0:
                    break
                m.add_rows(raw[0], raw[1], n)
                keys += E.plan.unpack_svector(K.T_STRING, raw[0])
                datas += E.plan.unpack_svector(K.T_STRING, raw[1])
            frames.append(O.partial_frame(keys, datas))
            q.close()
        got = m.fetch_all()
        assert got.nrows == exp.nrows
        T.compare_results(got.rows(), exp.rows(), exp.types, key_cols=nkeys)
        om = O.oracle_merge(whole, frames)
        T.compare_results(got.rows(), om.rows(), om.types, key_cols=nkeys, rel=0)
        m.close()


@pytest.mark.parametrize("bits", [1, 2, 3, 4, 5, 7, 8, 9, 11, 13, 16, 17, 21, 24, 27, 31, 32])
def test_every_bit_width_in_the_fused_kernel(ctx, bits):
    """UINT32_BITPACKED pages of every width class (word-aligned, straddling, full)
    read by the pairwise in-kernel decoder (evql_bitpacked_x2) as key, predicate
    operand and aggregate argument; 300,001 rows = several blocks past a page"""
    n = 300_001
    rng = np.random.default_rng(bits)
    maxv = (1 << bits) - 1
    x = rng.integers(0, maxv + 1, n, dtype=np.uint64)
    y = rng.integers(0, maxv + 1, n, dtype=np.uint64)
    w = E.Writer([dict(name="x", logical_type=K.COL_UNSIGNED_INT,
                       storage_type=K.ENC_UINT32_BITPACKED, bitpack_max_value=maxv),
                  dict(name="y", logical_type=K.COL_UNSIGNED_INT,
                       storage_type=K.ENC_UINT32_BITPACKED, bitpack_max_value=maxv)])
    w.put("x", x)
    w.put("y", y)
    w.commit(n)
    img = w.image()
    w.close()
    t = ctx.open_image(img)
    S = dict(x=K.T_UINT64, y=K.T_UINT64)
    plan = Plan(S, select=[col("x") % 13, count(1), sum_(col("y")), max_(col("x"))],
                group_by=[col("x") % 13], where=col("y") >= (maxv // 3))
    exp = O.oracle_run(img, plan)
    q = t.query(plan)
    T.compare_results(q.run().rows(), exp.rows(), exp.types)
    q.close()
    got = t.query(Plan(S, select=[sum_(col("x")), sum_(col("y"))])).run().rows()
    assert got == [(int(x.sum()), int(y.sum()))]
    t.close()


def test_operators_release_their_device_memory(ctx):
    """create / run / destroy many operators of every kind (LDS path, partitioned
    path, pair sets, ORDER BY, nested, string predicates) on one table: HBM in use
    returns to where it was (tables keep their decoded-column caches, so the
    baseline is taken after one warm-up round)"""
    import torch
    from eventql_amd.plan import count_distinct, Order
    img, _ = T.mixed_table(300_000)
    t = ctx.open_image(img)
    k, a, w, s = col("k"), col("a"), col("w"), col("s")
    plans = [
        (dict(select=[k, count(1), sum_(a)], group_by=[k], where=W), None),
        (dict(select=[w, count(1), sum_(a)], group_by=[w], groups_hint=3_000_000), None),
        (dict(select=[k, count_distinct(a)], group_by=[k]), None),
        (dict(select=[k, count(1), sum_(a)], group_by=[k]), [(2, True)]),
        (dict(select=[s, count(1)], group_by=[s], where=s.neq("g7")), None),
        (dict(select=[count(1), sum_(col("v"))]), None),
    ]

    def one_round():
        for kw, order in plans:
            plan = Plan(T.MIXED_SCHEMA, **kw)
            q = t.query(plan)
            if order:
                q.set_order(Order(plan, order, limit=5))
            q.run()
            q.close()

    one_round()
    ctx.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(25):
        one_round()
    ctx.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (64 << 20), (free0, free1)
    t.close()


def test_count_distinct(mixed):
    """count_distinct#uint64/uint64; (aggregate.cc:77-137): exact, through the HBM
    pair set, under every key mode, next to other aggregates"""
    from eventql_amd.plan import count_distinct as cd
    t, img, _ = mixed
    k, a, b, n, nb, w = col("k"), col("a"), col("b"), col("n"), col("nb"), col("w")
    check(t, img, select=[k, cd(a), count(1), cd(b % 7), sum_(a)], group_by=[k])
    check(t, img, select=[k, cd(nb), cd(n)], group_by=[k], where=W)     # NULL payloads read 0
    check(t, img, key_cols=0, select=[cd(a), cd(k), cd(w), count(1)])   # no GROUP BY: 300k distinct w
    check(t, img, key_cols=0, select=[cd(a)], where=a > 70000)          # no rows
    check(t, img, select=[nb, cd(a)], group_by=[nb])                    # NULL key group
    check(t, img, key_cols=2, select=[k, col("f"), cd(b)], group_by=[k, col("f")])   # hashed identity
    check(t, img, select=[col("s"), cd(a % 100)], group_by=[col("s")])  # string key
    check(t, img, select=[b, cd(k)], group_by=[b], groups_hint=70000)   # 65,536 groups
    check(t, img, select=[k, cd(w) + count(1)], group_by=[k])           # post-aggregate arithmetic
    # (the sets travel as pairs: test_count_distinct_through_export_and_import)
    # PartialGroupBy rows carry the set itself: varuint size, values ascending
    # (aggregate.cc:111-117) -- byte for byte the oracle's (pinned on the reference's
    # bytes in test_gpu_ref_csql.py)
    for kw in (dict(select=[k, cd(a), count(1)], group_by=[k]),
               dict(select=[cd(b % 97), cd(k)], group_by=[]),
               dict(select=[nb, cd(a)], group_by=[nb]),
               dict(select=[k, col("f"), cd(b)], group_by=[k, col("f")]),
               dict(select=[col("s"), cd(a % 100)], group_by=[col("s")])):
        plan = Plan(T.MIXED_SCHEMA, mode=K.MODE_PARTIAL, **kw)
        exp = O.oracle_run(img, plan)
        e = {exp.keys[20 * i:20 * i + 20]: exp.columns[0][i] for i in range(exp.nrows)}
        q = t.query(plan)
        g = dict(q.run().rows())
        q.close()
        assert g == e


def test_count_distinct_pair_set_regrows(ctx):
    """more distinct (group, value) pairs than the pair set starts with (2^20): the
    set is grown x4 and the query re-run"""
    n = 1_400_000
    w = E.Writer([dict(name="g", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
                  dict(name="x", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN)])
    i = np.arange(n, dtype=np.uint64)
    w.put("g", i % np.uint64(7))
    w.put("x", i * np.uint64(2654435761))
    w.commit(n)
    img = w.image()
    w.close()
    S = dict(g=K.T_UINT64, x=K.T_UINT64)
    plan = Plan(S, select=[col("g"), Agg("count_distinct", col("x")), count(1)], group_by=[col("g")])
    t = ctx.open_image(img)
    q = t.query(plan)
    exp = O.oracle_run(img, plan)
    T.compare_results(q.run().rows(), exp.rows(), exp.types)
    q.close()
    t.close()


def test_order_by_limit_above_the_group_by(mixed):
    """OrderByExpression + LimitExpression fused into the operator
    (evql_query_set_order): the device radix-selects the offset+limit records by the
    first sort key, the host orders them; vs the oracle's OrderBy/Limit restatement"""
    from eventql_amd.plan import Order, out
    t, img, _ = mixed
    k, a, v, nb, w = col("k"), col("a"), col("v"), col("nb"), col("w")
    sel = [k, count(1), sum_(a), mean(v), min_(nb), max_(col("p"))]

    def run(kw, specs, limit=None, offset=0, exact=True, keycols=None):
        plan = Plan(T.MIXED_SCHEMA, **kw)
        order = Order(plan, specs, limit=limit, offset=offset)
        exp = O.oracle_run(img, plan, order=order)
        q = t.query(plan)
        q.set_order(order)
        got = q.run()
        assert got.nrows == exp.nrows, (got.nrows, exp.nrows)
        assert q.stats()["num_groups"] == O.oracle_run(img, plan).nrows
        if exact:   # total order: row by row, floats within 1e-6
            for gr, er in zip(got.rows(), exp.rows()):
                T.compare_results([gr], [er], exp.types, key_cols=1)
        else:       # ties in unspecified order: the sort key columns agree row by row
            for gr, er in zip(got.rows(), exp.rows()):
                assert [gr[c] for c in keycols] == [er[c] for c in keycols]
        q.close()
        return got

    g = dict(select=sel, group_by=[k])
    got = run(g, [(2, True)], limit=10)                       # sum(a) desc
    assert [r[2] for r in got.rows()] == sorted((r[2] for r in got.rows()), reverse=True)
    run(g, [(1, False), (0, True)], limit=20, offset=5)       # count asc, k desc: total order
    run(g, [(0, False)], limit=7)                             # the group key itself
    run(g, [(0, True)], limit=7, offset=990)                  # runs off the end
    run(g, [(3, True)], limit=3)                              # mean(v): float key
    run(g, [(4, False), (0, False)], limit=15)                # min over a nullable: NULL reads 0
    run(g, [(5, True), (0, False)], limit=4)                  # max over a UINT32_PLAIN column
    run(g, [(1, True)], limit=25, exact=False, keycols=[1])   # ties at the cut
    run(g, [(2, False), (0, False)])                          # ORDER BY without LIMIT
    run(g, [(2, False)], limit=0)
    run(g, [(2, False), (0, True)], limit=5000)               # limit > groups
    run(g, [(0, False), (out(2) + out(1), True)], limit=9)    # expression as a later key
    assert run(g, [], limit=13, exact=False, keycols=[]).nrows == 13   # LIMIT alone
    run(dict(select=[count(1), sum_(a)]), [(0, False)], limit=1)       # no GROUP BY

    # high cardinality: 65,536 and ~300,000 groups, 100 leave the device
    hc = dict(select=[col("b"), count(1), sum_(a), max_(v)], group_by=[col("b")])
    run(hc, [(2, True), (0, False)], limit=100)
    run(hc, [(3, False), (0, False)], limit=100, offset=40000)
    run(dict(select=[w, sum_(a)], group_by=[w], groups_hint=400000), [(1, True), (0, True)],
        limit=50)
    run(dict(select=[w, count(1)], group_by=[w], groups_hint=400000), [(1, True), (0, False)],
        limit=50)                                             # all ties on key 0: full tie set

    # first sort key not readable from a group record -> the CPU operators stay above
    plan = Plan(T.MIXED_SCHEMA, **g)
    for specs in ([(out(2) + out(1), True)], [(out(1) * 2, False)]):
        q = t.query(plan)
        with pytest.raises(E.EvqlError) as ei:
            q.set_order(Order(plan, specs, limit=10))
        assert ei.value.code == K.EVQL_ENOTSUP
        q.close()
    plan = Plan(T.MIXED_SCHEMA, select=[col("s"), count(1)], group_by=[col("s")])
    q = t.query(plan)
    with pytest.raises(E.EvqlError) as ei:
        q.set_order(Order(plan, [(0, False)], limit=10))     # string key: identity is a hash
    assert ei.value.code == K.EVQL_ENOTSUP
    q.set_order(Order(plan, [(1, True), (0, False)], limit=10))   # .. but fine as 2nd key
    exp = O.oracle_run(img, plan, order=Order(plan, [(1, True), (0, False)], limit=10))
    assert q.run().rows() == exp.rows()
    q.close()


def test_nested_scan_known_answers(ctx):
    """Dremel flattening (CSTableScan, NO_AGGREGATION) on the device: the
    Runtime_test.cc:175-375 answers on the reference's fixture (re-encoded as
    v0.2.0), each also compared with the oracle"""
    import nested_tables as N
    img = N.testtbl_v2()
    t = ctx.open_image(img)
    S = N.NESTED_SCHEMA
    tm = col("time")
    sq_time = col("event.search_query.time")
    nitems = col("event.search_query.num_result_items")
    pos = col("event.search_query.result_items.position")
    clicked = col("event.search_query.result_items.clicked")
    cases = [
        (dict(select=[count(1)]), [(213,)]),                             # no columns: per record
        # fetchNextWithoutColumns skips a record when WHERE is TRUE (CSTableScan.cc:551-564)
        (dict(select=[count(1)], where=lit(1) > lit(2)), [(213,)]),
        (dict(select=[count(1)], where=lit(2) > lit(1)), []),
        (dict(select=[count(sq_time)]), [(773,)]),                       # 704 defined + 69 empty
        (dict(select=[count(1)], where=sq_time > 0), [(704,)]),
        (dict(select=[sum_(nitems)]), [(24793,)]),
        (dict(select=[count(1), sum_(If(clicked, 1, 0))], where=pos.eq(6)), [(688, 2)]),
        # parents repeat per leaf slot: 24866 flattened rows
        (dict(select=[count(tm), count(sq_time), sum_(nitems), count(pos)]), None),
        (dict(select=[pos, count(1), sum_(nitems), max_(tm)], group_by=[pos]), None),
        (dict(select=[nitems, count(1), sum_(pos)], group_by=[nitems]), None),
    ]
    for kw, known in cases:
        plan = Plan(S, scan_mode=K.SCAN_NESTED, **kw)
        exp = O.oracle_run(img, plan)
        if known is not None:
            assert exp.rows() == known
        q = t.query(plan)
        got = q.run()
        T.compare_results(got.rows(), exp.rows(), exp.types, key_cols=len(kw.get("group_by", [])))
        assert q.stats()["rows_passed"] == exp.rows_passed
        q.close()
    # WHERE mixing repetition depths: after a rejected row the reference resets parent
    # values without re-reading them (CSTableScan.cc:501-512) -- reproduced; the
    # reference-generated cases are in test_gpu_ref_csql.py
    for where in ((pos > 3) & (nitems > 10), (nitems > 10) | (pos < 2),
                  ~(col("time") > 1438055327000000) | (pos > 7)):
        plan = Plan(S, select=[nitems, count(1), sum_(pos), sum_(col("time"))], group_by=[nitems],
                    where=where, scan_mode=K.SCAN_NESTED)
        exp = O.oracle_run(img, plan)
        q = t.query(plan)
        assert "evql_where_rows" in q.kernel_source()
        T.compare_results(q.run().rows(), exp.rows(), exp.types, key_cols=1)
        q.close()
    t.close()


def test_v010_fixture_opened_directly(ctx):
    """the reference's own fixture file (cstable v0.1.0) through
    evql_table_open_file: test/sql/00002 (count), 00001 (the time column, via a
    group by) and the Runtime_test.cc nested answers, oracle reading the v0.1.0
    file itself"""
    import nested_tables as N
    import os
    path = os.path.join(T.GOLDEN, "testtbl.cst")
    t = ctx.open_file(path)
    assert t.num_rows == 213
    S = N.NESTED_SCHEMA
    tm = col("time")
    nitems = col("event.search_query.num_result_items")
    pos = col("event.search_query.result_items.position")
    clicked = col("event.search_query.result_items.clicked")
    got = t.query(Plan(S, select=[count(1)])).run()
    assert got.rows() == [(213,)]
    exp_times = [int(x) for x in open(os.path.join(
        T.GOLDEN, "00001_test_column_reference_with_table_name_prefix.result.txt")
    ).read().split("\n")[1:] if x]
    got = t.query(Plan(S, select=[tm, count(1)], group_by=[tm])).run()
    from collections import Counter
    assert dict(got.rows()) == dict(Counter(exp_times))
    for kw, known in [
            (dict(select=[sum_(nitems)]), [(24793,)]),
            (dict(select=[count(1), sum_(If(clicked, 1, 0))], where=pos.eq(6)), [(688, 2)]),
            (dict(select=[pos, count(1), sum_(nitems), max_(tm)], group_by=[pos]), None)]:
        plan = Plan(S, scan_mode=K.SCAN_NESTED, **kw)
        exp = O.oracle_run(path, plan)
        if known is not None:
            assert exp.rows() == known
        q = t.query(plan)
        T.compare_results(q.run().rows(), exp.rows(), exp.types,
                          key_cols=len(kw.get("group_by", [])))
        q.close()
    t.close()


def test_nested_scan_synthetic_items(ctx):
    """config-5 shape: REPEATED RECORD items{position, price}, 0..8 per record"""
    import nested_tables as N
    img, st = N.items_table(100_000)
    t = ctx.open_image(img)
    S = N.ITEMS_SCHEMA
    pos, price, rid, score = col("items.position"), col("items.price"), col("id"), col("score")
    for kw in (dict(select=[count(1), sum_(price), sum_(pos)]),
               dict(select=[count(1), sum_(price)], where=pos > 0),
               dict(select=[pos, count(1), sum_(price), min_(price), max_(price)], group_by=[pos]),
               dict(select=[pos, count(1), sum_(rid), sum_(score)], group_by=[pos]),
               dict(select=[count(1), sum_(rid)])):
        plan = Plan(S, scan_mode=K.SCAN_NESTED, **kw)
        exp = O.oracle_run(img, plan)
        q = t.query(plan)
        got = q.run()
        T.compare_results(got.rows(), exp.rows(), exp.types, key_cols=len(kw.get("group_by", [])))
        q.close()
    got = t.query(Plan(S, select=[count(1), sum_(price), sum_(pos)], scan_mode=K.SCAN_NESTED)).run()
    assert got.rows() == [(st["total"], st["sum_price"], st["sum_pos"])]
    t.close()


def test_within_record_scan(ctx):
    """CSTableScan's AGGREGATE_WITHIN_RECORD_FLAT (CSTableScan.cc:440-487): one row
    per record of per-record count / sum, consumed by the group-by above it --
    `select sum(count(x) WITHIN RECORD)` and friends on the reference's fixture
    (v0.1.0 file and its v0.2.0 re-encoding) and on the synthetic items table"""
    import nested_tables as N
    import os
    from eventql_amd.plan import out
    WR = K.SCAN_NESTED_WITHIN_RECORD
    S = N.NESTED_SCHEMA
    tm = col("time")
    sq_time = col("event.search_query.time")
    nitems = col("event.search_query.num_result_items")
    pos = col("event.search_query.result_items.position")
    clicked = col("event.search_query.result_items.clicked")
    cases = [
        (dict(scan_select=[count(sq_time), sum_(nitems), count(pos), count(1)],
              select=[sum_(out(0)), sum_(out(1)), sum_(out(2)), sum_(out(3)), count(1)]),
         [(773, 24793, 24866, 213, 213)]),
        # per-record item counts as the group key
        (dict(scan_select=[count(pos), sum_(nitems), sum_(pos), count(clicked)],
              select=[out(0), count(1), sum_(out(1)), max_(out(2)), min_(out(3))],
              group_by=[out(0)]), None),
        (dict(scan_select=[count(sq_time), sum_(tm), sum_(lit(2))],
              select=[out(0), count(1), sum_(out(1)), sum_(out(2))], group_by=[out(0)]), None),
        (dict(scan_select=[sum_(tm), count(1)], select=[sum_(out(0)), sum_(out(1))]), None),
        # non-aggregate select expressions read the group's first record
        (dict(scan_select=[count(sq_time), sum_(nitems), sum_(tm)],
              select=[out(0), out(1) + 1, out(2), count(1)], group_by=[out(0)]), None),
    ]
    for img in (N.testtbl_v2(), os.path.join(T.GOLDEN, "testtbl.cst")):
        t = ctx.open_file(img) if isinstance(img, str) else ctx.open_image(img)
        for kw, known in cases:
            plan = Plan(S, scan_mode=WR, **kw)
            exp = O.oracle_run(img, plan)
            if known is not None:
                assert exp.rows() == known
            q = t.query(plan)
            got = q.run()
            T.compare_results(got.rows(), exp.rows(), exp.types,
                              key_cols=len(kw.get("group_by", [])))
            assert q.stats()["rows_passed"] == 213
            q.close()
        t.close()
    img, st = N.items_table(100_000)
    t = ctx.open_image(img)
    S = N.ITEMS_SCHEMA
    ipos, price, rid = col("items.position"), col("items.price"), col("id")
    for kw in (dict(scan_select=[count(ipos), sum_(price)],
                    select=[out(0), count(1), sum_(out(1)), max_(out(1))], group_by=[out(0)]),
               dict(scan_select=[sum_(price), sum_(ipos), count(1), sum_(rid)],
                    select=[sum_(out(0)), sum_(out(1)), sum_(out(2)), sum_(out(3))]),
               dict(scan_select=[sum_(rid), count(rid)],
                    select=[out(0) % 5, sum_(out(1))], group_by=[out(0) % 5])):
        plan = Plan(S, scan_mode=WR, **kw)
        exp = O.oracle_run(img, plan)
        q = t.query(plan)
        T.compare_results(q.run().rows(), exp.rows(), exp.types,
                          key_cols=len(kw.get("group_by", [])))
        q.close()
    got = t.query(Plan(S, scan_select=[sum_(price), sum_(ipos)],
                       select=[sum_(out(0)), sum_(out(1)), count(1)], scan_mode=WR)).run()
    assert got.rows() == [(st["sum_price"], st["sum_pos"], 100_000)]
    # a record scan without columns is undefined in the reference (null instance)
    with pytest.raises(E.EvqlError) as ei:
        t.query(Plan(S, scan_select=[count(1)], select=[sum_(out(0))], scan_mode=WR))
    assert ei.value.code == K.EVQL_EARG
    # not lowered: WHERE, expressions inside the per-record aggregate
    for bad in (dict(scan_select=[count(1)], select=[sum_(out(0))], where=rid > 5),
                dict(scan_select=[sum_(price * 2)], select=[sum_(out(0))])):
        with pytest.raises(E.EvqlError) as ei:
            t.query(Plan(S, scan_mode=WR, **bad))
        assert ei.value.code == K.EVQL_ENOTSUP
    t.close()


def test_within_record_long_records(ctx):
    """records of 0 .. 9000 items: spanning many 8-row windows and 2048-row tiles of
    k_within_record (plain stores inside a tile, atomics across tile borders),
    against numpy segment sums and the oracle"""
    from eventql_amd.plan import out
    rng = np.random.default_rng(11)
    cnt = rng.choice([0, 1, 2, 7, 8, 9, 40, 300, 2047, 2048, 2049, 5000, 9000], 400)
    nrec = len(cnt)
    slots = np.maximum(cnt, 1)
    total = int(slots.sum())
    starts = np.concatenate([[0], np.cumsum(slots)[:-1]])
    rl = np.ones(total, np.uint64)
    rl[starts] = 0
    rec_of_slot = np.repeat(np.arange(nrec), slots)
    dl = np.where(cnt[rec_of_slot] > 0, 2, 0).astype(np.uint64)
    price = rng.integers(1, 1 << 40, total).astype(np.uint64)
    w = E.Writer([
        dict(name="id", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
        dict(name="items.price", logical_type=K.COL_UNSIGNED_INT,
             storage_type=K.ENC_UINT64_PLAIN, rlevel_max=1, dlevel_max=2)])
    w.put("id", np.arange(nrec, dtype=np.uint64))
    w.put("items.price", price, rlvl=rl, dlvl=dl)
    w.commit(nrec)
    img = w.image()
    w.close()
    t = ctx.open_image(img)
    S = {"id": K.T_UINT64, "items.price": K.T_UINT64}
    plan = Plan(S, scan_select=[sum_(col("id")), count(col("items.price")),
                                sum_(col("items.price")), sum_(lit(3))],
                select=[out(0), sum_(out(1)), sum_(out(2)), sum_(out(3)), count(1)],
                group_by=[out(0)], scan_mode=K.SCAN_NESTED_WITHIN_RECORD)
    got = sorted(t.query(plan).run().rows())
    defined = dl == 2
    psum = np.bincount(rec_of_slot, weights=None, minlength=nrec)
    exp = []
    for r in range(nrec):
        seg = slice(starts[r], starts[r] + slots[r])
        # (a literal has repetition level 0: accumulated on the record's first row only)
        exp.append((r, int(slots[r]), int(price[seg][defined[seg]].sum(dtype=np.uint64)), 3, 1))
    assert psum.tolist() == slots.tolist()
    assert got == exp
    T.compare_results(got, sorted(O.oracle_run(img, plan).rows()), [K.T_UINT64] * 5, key_cols=1)
    t.close()


def _small_table(ctx, cols, specs, n):
    w = E.Writer(specs)
    for s in specs:
        w.put(s["name"], cols[s["name"]], present=cols.get(s["name"] + "_present"))
    w.commit(n)
    img = w.image()
    w.close()
    return ctx.open_image(img), img


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 2047, 2048, 2049, 65536, 65537, 131073])
def test_edge_sizes(ctx, n):
    c = synth.table_columns(max(n, 1))
    cols = {k: v[:n] for k, v in c.items()}
    specs = [dict(name="k", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT32_BITPACKED,
                  bitpack_max_value=1023),
             dict(name="a", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
             dict(name="v", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754),
             dict(name="b", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_LEB128)]
    t, img = _small_table(ctx, cols, specs, n)
    S = dict(k=K.T_UINT64, a=K.T_UINT64, v=K.T_FLOAT64, b=K.T_UINT64)
    for kw in (dict(select=[col("k"), count(1), sum_(col("a")), sum_(col("v")), sum_(col("b"))],
                    group_by=[col("k")]),
               dict(select=[count(1), sum_(col("b"))], where=col("a") > 30000)):
        plan = Plan(S, **kw)
        exp = O.oracle_run(img, plan)
        got = t.query(plan).run()
        assert got.nrows == exp.nrows
        T.compare_results(got.rows(), exp.rows(), exp.types, key_cols=len(kw.get("group_by", [])))
    t.close()


def test_sentinel_and_extreme_keys(ctx):
    """the key value 2^64-1 is the table's EMPTY marker internally; it must
    still come out as an ordinary group, next to NULL and 0"""
    n = 5000
    i = np.arange(n, dtype=np.uint64)
    key = np.where(i % 5 == 0, np.uint64(0xFFFFFFFFFFFFFFFF),
                   np.where(i % 5 == 1, np.uint64(0), i % np.uint64(7)))
    cols = dict(key=key, key_present=(i % 11 != 3).astype(np.uint8), a=i)
    specs = [dict(name="key", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN,
                  dlevel_max=1),
             dict(name="a", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN)]
    t, img = _small_table(ctx, cols, specs, n)
    S = dict(key=K.T_UINT64, a=K.T_UINT64)
    for hint in (0, 100000):
        plan = Plan(S, select=[col("key"), count(1), sum_(col("a")), max_(col("a"))],
                    group_by=[col("key")], groups_hint=hint)
        exp = O.oracle_run(img, plan)
        got = t.query(plan).run()
        T.compare_results(got.rows(), exp.rows(), exp.types)
        keys = [r[0] for r in got.rows()]
        assert 0xFFFFFFFFFFFFFFFF in keys and None in keys and 0 in keys
    t.close()


def test_survey_goldens_on_gpu(ctx):
    """the reference outputs of SURVEY.md 8c(ii), straight from the HIP path"""
    gold = json.load(open(os.path.join(T.GOLDEN, "survey_8c.json")))
    img, _ = T.survey_table(1_000_000)
    t = ctx.open_image(img)

    def run(**kw):
        return t.query(Plan(T.SURVEY_SCHEMA, **kw)).run()

    assert run(select=[count(1)]).rows() == [(gold["count_1"],)]
    assert run(select=[count(1)], where=W).rows() == \
        [(gold["count_where_a_gt_30000_and_b_lt_30000"],)]
    d = {r[0]: r for r in run(select=[col("k"), sum_(col("a")), count(1), sum_(col("b"))],
                              group_by=[col("k")], where=W).rows()}
    g0, g1 = gold["filtered_k0"], gold["filtered_k1"]
    assert d[0] == (0, g0["sum_a"], g0["count"], g0["sum_b"])
    assert d[1] == (1, g1["sum_a"], g1["count"], g1["sum_b"])
    assert run(select=[sum_(col("n")), count(col("n"))], where=col("n") >= 0).rows() == \
        [tuple(gold["sum_n_count_n_where_n_gte_0"])]
    assert run(select=[count(1)], where=col("n") > 5).rows() == [(gold["count_where_n_gt_5"],)]
    assert run(select=[count(1)], where=~((col("a") > 30000) | col("b").eq(5))).rows() == \
        [(gold["count_where_not_a_gt_30000_or_b_eq_5"],)]
    d = {r[0]: r[1] for r in run(select=[col("s"), count(1)], group_by=[col("s")]).rows()}
    for k, v in gold["string_groups"].items():
        assert d[k.encode()] == v
    r = run(select=[col("n"), count(1)], group_by=[col("n")], groups_hint=700000)
    assert r.nrows == gold["high_cardinality_groups_n"]
    assert {x[0]: x[1] for x in r.rows()}[None] == gold["null_group_count"]
    assert run(select=[sum_(col("b") * 281474976710656)]).rows() == [(gold["sum_b_times_2_48"],)]
    t.close()


def test_device_generator_matches_host_twin(ctx, tmp_path):
    n = 700_001
    t = ctx.generate(n, "kabvu", u_mod=10_000_000)
    img = t.download_image()
    path = str(tmp_path / "gen.cst")
    open(path, "wb").write(img)
    c = synth.table_columns(n)
    rd = O.TableReader(path, "orc")
    for name in "kab":
        assert (rd.read(name, n, "uint")[3] == c[name]).all()
    assert (rd.read("v", n, "float")[3] == c["v"]).all()
    assert (rd.read("u", n, "uint")[3] == c["x"] % np.uint64(10_000_000)).all()
    rd.close()
    if O.have_ref():
        rr = O.TableReader(path, "ref")
        assert rr.num_rows == n and (rr.read("k", n, "uint")[3] == c["k"]).all()
        rr.close()
    t.close()
    # bit-packed key column at 10 bits, as in config 2 run B
    t = ctx.generate(300_000, "kv", k_bits=10)
    img = t.download_image()
    open(path, "wb").write(img)
    rd = O.TableReader(path, "orc")
    assert (rd.read("k", 300_000, "uint")[3] == c["k"][:300_000]).all()
    rd.close()
    p = B.config2()
    exp = O.oracle_run(path, p)
    got = t.query(p).run()
    T.compare_results(got.rows(), exp.rows(), exp.types)
    t.close()


def test_full_size_high_cardinality_properties(ctx):
    """BASELINE config 4 at full size (1.25e8 rows, key uniform in [0, 1e7)): the
    partitioned path against totals, against itself, and -- group by group on the
    100,000 smallest keys -- against the LDS path of the same query with a WHERE"""
    from eventql_amd.plan import Order
    n, n_keys = 125_000_000, 10_000_000
    t = ctx.generate(n, "uav", u_mod=n_keys)
    plan = B.config4(groups_hint=n_keys)
    q = t.query(plan)
    assert "evql_part_refine" in q.kernel_source()
    q.set_order(Order(plan, [(0, False)], limit=100_000))      # 100,000 smallest keys
    head = q.run()
    st = q.stats()
    assert head.nrows == 100_000
    assert n_keys - 200 < st["num_groups"] <= n_keys and st["rows_passed"] == n
    keys = [r[0] for r in head.rows()]
    assert keys == sorted(keys) and len(set(keys)) == len(keys) and keys[-1] < 100_500
    q.close()
    # totals through plain global aggregates (register accumulators)
    tot = t.query(Plan(B.SCHEMA, select=[count(1), sum_(col("a")), sum_(col("v"))])).run().rows()[0]
    # ... equal the sums over all groups, read through a second top-k ordering
    q = t.query(plan)
    q.set_order(Order(plan, [(2, True), (0, False)], limit=50))  # largest counts
    top = q.run().rows()
    assert [r[2] for r in top] == sorted((r[2] for r in top), reverse=True)
    q.close()
    q = t.query(plan)
    q.execute()
    allg = q.fetch_all(1 << 20)       # 1e7 rows through nextBatch
    assert allg.nrows == st["num_groups"]
    assert sum(r[2] for r in allg.rows()) == n == tot[0]
    assert sum(r[1] for r in allg.rows()) == tot[1]
    sv = sum(r[3] for r in allg.rows())
    assert abs(sv - tot[2]) <= 1e-9 * abs(tot[2])
    by_key = {r[0]: r for r in allg.rows()}
    assert max(r[2] for r in allg.rows()) == top[0][2]
    q.close()
    # the same groups through a different code path: WHERE u < 100,000 -> ~1e5
    # groups, LDS / HBM-atomic path
    q = t.query(Plan(B.SCHEMA, select=[col("u"), sum_(col("a")), count(1), sum_(col("v"))],
                     group_by=[col("u")], where=col("u") < 100_000, groups_hint=100_000))
    assert "evql_part_refine" not in q.kernel_source()
    small = q.run()
    assert small.nrows == sum(1 for k in by_key if k < 100_000)
    for k, sa, c, v in small.rows():
        r = by_key[k]
        assert (sa, c) == (r[1], r[2]), k
        assert abs(v - r[3]) <= 1e-9 * abs(v), k
    for r_head in head.rows():
        assert r_head[1:3] == by_key[r_head[0]][1:3]
    q.close()
    t.close()


def test_full_size_string_key_properties(ctx):
    """BASELINE config 4 as written -- 1.25e8 rows, 1e7 STRING keys ("g" + u): value
    boundaries and hashes found on the device, partitioned path with 32-byte tuples,
    first-row strings gathered at emission -- against totals and, group by group on a
    1111-key slice, against the LDS path of the same query with a WHERE"""
    n, n_keys = 125_000_000, 10_000_000
    t = B.string_key_table(ctx, n, n_keys, seed=77)
    S = B.STRING_KEY_SCHEMA
    plan = B.config4s(groups_hint=n_keys)
    q = t.query(plan)
    assert "evql_part_refine" in q.kernel_source()
    q.execute()
    st = q.stats()
    assert n_keys - 200 < st["num_groups"] <= n_keys and st["rows_passed"] == n
    allg = q.fetch_all(1 << 20)
    q.close()
    assert allg.nrows == st["num_groups"]
    rows = allg.rows()
    by_key = {r[0]: r for r in rows}
    assert len(by_key) == allg.nrows                      # every key once
    assert all(k[:1] == b"g" for k in list(by_key)[:1000])
    tot = t.query(Plan(S, select=[count(1), sum_(col("a")), sum_(col("v"))])).run().rows()[0]
    assert sum(r[2] for r in rows) == n == tot[0]
    assert sum(r[1] for r in rows) == tot[1]
    sv = sum(r[3] for r in rows)
    assert abs(sv - tot[2]) <= 1e-9 * abs(tot[2])
    # keys g9999, g9999x, g9999xx, g9999xxx through the LDS path (bytewise predicates)
    s_ = col("s")
    q = t.query(Plan(S, select=[s_, sum_(col("a")), count(1), sum_(col("v"))], group_by=[s_],
                     where=(s_ >= "g9999") & (s_ < "g9999:"), groups_hint=2000))
    assert "evql_part_refine" not in q.kernel_source()
    small = q.run()
    q.close()
    want = [k for k in by_key if k.startswith(b"g9999")]
    assert small.nrows == len(want) and 1000 < small.nrows <= 1111
    for k, sa, c, v in small.rows():
        r = by_key[k]
        assert (sa, c) == (r[1], r[2]), k
        assert abs(v - r[3]) <= 1e-9 * abs(v), k
    t.close()


def test_full_size_properties(ctx):
    """BASELINE sizes (1e9 rows, config 3): size-independent properties"""
    n = 1_000_000_000
    t = ctx.generate(n, "kabv")
    q = t.query(B.config3())
    r1 = q.run()
    st = q.stats()
    assert r1.nrows == 1000
    # checksum of checksums: the per-group counts add up to the passing rows
    assert sum(r[2] for r in r1.rows()) == st["rows_passed"]
    assert 0.24 * n < st["rows_passed"] < 0.26 * n
    # idempotence: integer aggregates identical run to run, float sums within 1e-9
    r2 = q.run()
    d1 = {r[0]: r for r in r1.rows()}
    for r in r2.rows():
        assert r[2] == d1[r[0]][2] and r[3] == d1[r[0]][3]
        assert abs(r[1] - d1[r[0]][1]) <= 1e-9 * abs(r[1])
    q.close()
    # linearity: two half-table slices add up to the whole
    qa = t.query(B.config3(row_end=n // 2 + 12345))
    qb = t.query(B.config3(row_begin=n // 2 + 12345))
    da = {r[0]: r for r in qa.run().rows()}
    db = {r[0]: r for r in qb.run().rows()}
    for k, r in d1.items():
        assert da[k][2] + db[k][2] == r[2] and da[k][3] + db[k][3] == r[3]
        assert abs(da[k][1] + db[k][1] - r[1]) <= 1e-9 * abs(r[1])
    # the 1M-row prefix equals the host-generated table through the oracle
    qp = t.query(B.config3(row_end=1_000_000))
    got = qp.run()
    c = synth.table_columns(1_000_000)
    m = (c["a"] > 30000) & (c["b"] < 30000)
    for k, sv, cnt, sb in got.rows():
        mk = m & (c["k"] == k)
        assert cnt == int(mk.sum()) and sb == int(c["b"][mk].sum())
        assert abs(sv - float(c["v"][mk].sum())) <= 1e-6 * abs(sv)
    for x in (qa, qb, qp):
        x.close()
    # count(1) over everything, no predicate
    q = t.query(Plan(B.SCHEMA, select=[count(1)]))
    assert q.run().rows() == [(n,)]
    q.close()
    t.close()


def test_plan_without_a_hint_finds_the_partitioned_path_by_itself(ctx):
    """groups_hint = 0 (the reference's planner has no cardinality estimate): the
    first execute aggregates a 256 Ki-row prefix, estimates the number of groups and
    re-shapes the plan.  3e6 groups over 9e6 rows -> the partitioned kernels; 1000
    groups over the same rows -> the LDS path stays."""
    n = 9_000_000
    t = ctx.generate(n, "kuab", u_mod=3_000_000)
    try:
        img = t.download_image()
        S = dict(k=K.T_UINT64, u=K.T_UINT64, a=K.T_UINT64, b=K.T_UINT64)
        u, k, a, b = col("u"), col("k"), col("a"), col("b")
        plan = Plan(S, select=[u, count(1), sum_(a), max_(b)], group_by=[u], where=a > 1000)
        q = t.query(plan)
        assert "evql_part_scatter" not in q.kernel_source()
        got = q.run()
        st = q.stats()
        assert "evql_part_scatter" in q.kernel_source()
        assert 2_000_000 < st["estimated_groups"] < 5_000_000, st
        exp = O.oracle_run(img, plan)
        assert got.nrows == exp.nrows
        T.compare_results(got.rows(), exp.rows(), exp.types)
        # a second execute of the same operator does not probe again
        got2 = q.run()
        assert got2.nrows == exp.nrows
        q.close()
        plan = Plan(S, select=[k, count(1), sum_(a)], group_by=[k], where=a > 1000)
        q = t.query(plan)
        got = q.run()
        assert "evql_part_scatter" not in q.kernel_source()
        assert 900 < q.stats()["estimated_groups"] < 1500
        exp = O.oracle_run(img, plan)
        T.compare_results(got.rows(), exp.rows(), exp.types)
        q.close()
    finally:
        t.close()


def test_partitioned_path_with_a_dominant_key(ctx):
    """the fused two-level partitioning gives every coarse bucket a slack-allocated range
    (no count pass); half of the rows carrying ONE key overflow that key's range: the
    launch is void, the operator falls back to exact offsets (evql_part_count) and the
    result is the same"""
    n = 6_000_000
    rng = np.random.default_rng(77)
    u = rng.integers(0, 2_000_000, n).astype(np.uint64)
    u[rng.random(n) < 0.5] = 1234567
    a = rng.integers(0, 65536, n).astype(np.uint64)
    w = E.Writer([dict(name="u", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
                  dict(name="a", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN)])
    w.put("u", u)
    w.put("a", a)
    w.commit(n)
    img = w.image()
    w.close()
    t = ctx.open_image(img)
    try:
        S = dict(u=K.T_UINT64, a=K.T_UINT64)
        plan = Plan(S, select=[col("u"), count(1), sum_(col("a")), max_(col("a"))],
                    group_by=[col("u")], groups_hint=2_000_000)
        q = t.query(plan)
        assert "evql_part_refine" in q.kernel_source() and "evql_part_count(" not in q.kernel_source()
        got = q.run()
        assert "evql_part_count(" in q.kernel_source()       # fell back
        keys, cnt = np.unique(u, return_counts=True)
        assert got.nrows == len(keys)
        rows = {r[0]: r for r in got.rows()}
        assert rows[1234567][1] == int(cnt[keys == 1234567][0])
        assert rows[1234567][2] == int(a[u == 1234567].sum())
        sample = keys[:: max(1, len(keys) // 500)]
        for kk in sample:
            m = u == kk
            assert rows[int(kk)][1:] == (int(m.sum()), int(a[m].sum()), int(a[m].max()))
        # the same operator again: exact offsets right away
        assert q.run().nrows == len(keys)
        q.close()
    finally:
        t.close()


def test_partitioned_path_with_a_handful_of_keys(mixed):
    """a hint of 3e6 groups over rows that hold 3 .. 40 keys: every tuple lands in a few
    coarse buckets, which overflow their slack -- whichever they are, the last range of
    the buffer included.  The void launch must not read or write outside its buffers
    (found by the round-3 fuzz soak, seed 4155: evql_part_refine followed the overflowed
    cursor past the end of the tuple buffer) and the fallback gives the reference's rows"""
    t, img, _ = mixed
    a, k, nb, n = col("a"), col("k"), col("nb"), col("n")
    for key in ((nb * n + 1099511627776) % 3, k % 5 + 1000003, a % 40, (a % 7) * 2305843009213693951,
                k % 3 + 17, k % 2 + 123456789012):
        check(t, img, select=[key, mean(col("p") * 1000), max_(nb / (col("p") + 1)), count(1)],
              group_by=[key], groups_hint=3_000_000, row_end=131073)
        check(t, img, select=[key, count(1), sum_(a)], group_by=[key], groups_hint=3_000_000)


def test_hint_less_plan_over_keys_that_follow_the_row_order(ctx):
    """a time-ordered table: the key grows with the row number, so a prefix of the scan
    holds a handful of groups while the table has 90,000.  The probe samples row ranges
    spread over the whole scan range; a wrong estimate would still give the
    right rows (TABLE_FULL / DENSE_FULL re-runs), only slower"""
    n = 9_000_000
    i = np.arange(n, dtype=np.uint64)
    w = E.Writer([dict(name="ts", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
                  dict(name="a", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN)])
    w.put("ts", i // np.uint64(100))
    w.put("a", (i * np.uint64(2654435761)) % np.uint64(65536))
    w.commit(n)
    img = w.image()
    w.close()
    t = ctx.open_image(img)
    try:
        S = dict(ts=K.T_UINT64, a=K.T_UINT64)
        plan = Plan(S, select=[col("ts"), count(1), sum_(col("a"))], group_by=[col("ts")])
        q = t.query(plan)
        got = q.run()
        st = q.stats()
        # (within the probe's sample every key is seen ~100 times in 16 runs of consecutive
        # rows: the occupancy estimate sees 16 * 164 distinct keys among 262,144 rows and
        # cannot know better than "at least that many"; what matters is that it is no
        # longer the prefix's 2,622)
        assert st["estimated_groups"] >= 2_600, st
        assert got.nrows == 90_000
        rows = sorted(got.rows())
        assert [r[0] for r in rows] == list(range(90_000))
        assert all(r[1] == 100 for r in rows)
        a = (i * np.uint64(2654435761)) % np.uint64(65536)
        assert [r[2] for r in rows[:1000]] == a[:100_000].reshape(1000, 100).sum(axis=1).tolist()
        q.close()
    finally:
        t.close()


def test_exact_float_sums_refuse_mismatched_quanta(ctx):
    """with float_sum_bound = 0 every partition derives its quantum from its
    OWN maxima.  Two partitions whose maxima lie on either side of a power of two must
    not have their integer state words added: the exchange (and import_groups) answers
    EVQL_EARG instead of a silently wrong 'exact' sum; with one explicit bound the same
    exchange works"""
    import threading

    def table(scale):
        n = 50_000
        w = E.Writer([dict(name="k", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
                      dict(name="v", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754)])
        w.put("k", np.arange(n, dtype=np.uint64) % np.uint64(7))
        w.put("v", (np.arange(n, dtype=np.float64) % 1000.0) * scale)
        w.commit(n)
        img = w.image()
        w.close()
        return img
    imgs = [table(1.0), table(5.0)]  # maxima 999 and 4995: different binades
    S = dict(k=K.T_UINT64, v=K.T_FLOAT64)
    kw = dict(select=[col("k"), sum_(col("v"))], group_by=[col("k")])

    def run(bound):
        hub = E.Hub(2)
        out = [None, None]

        def work(r):
            cx = E.Context(0)
            tt = cx.open_image(imgs[r])
            qq = tt.query(Plan(S, float_sum_mode=K.FLOAT_SUM_EXACT, float_sum_bound=bound, **kw))
            x = E.Exchange.hub(cx, hub, r)
            qq.execute()
            try:
                qq.exchange(x, K.EXCHANGE_GATHER_ALL)
                out[r] = sorted(qq.fetch_all().rows())
            except E.EvqlError as e:
                out[r] = e
            qq.close(); x.close(); tt.close(); cx.close()
        th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
        [x.start() for x in th]
        [x.join(timeout=300) for x in th]
        hub.close()
        return out
    bad = run(0.0)
    assert all(isinstance(o, E.EvqlError) and o.code == K.EVQL_EARG and "quant" in o.msg for o in bad), bad
    good = run(8192.0)
    exp = {}
    for r, scale in enumerate((1.0, 5.0)):
        for i in range(50_000):
            exp[i % 7] = exp.get(i % 7, 0.0) + (i % 1000) * scale
    assert good[0] == good[1] == sorted(exp.items())
    # export / import: only with an explicit bound
    import torch
    t = ctx.open_image(imgs[0])
    q = t.query(Plan(S, float_sum_mode=K.FLOAT_SUM_EXACT, **kw))
    q.execute()
    buf = torch.zeros(1 << 12, dtype=torch.int64, device="cuda")
    with pytest.raises(E.EvqlError) as ei:
        q.export_groups(buf.data_ptr(), 64)
    assert ei.value.code == K.EVQL_EARG
    q.close()
    t.close()


def test_exact_float_sums(ctx, mixed):
    """EVQL_FLOAT_SUM_EXACT: every value is rounded once to a multiple of a power of two
    and the multiples are added as integers -- bit-stable from run to run and for any
    split of the rows; on this table (v = integer / 1024) even the exactly rounded sum"""
    import math
    import threading
    t, img, c = mixed
    k, v, nv, a = col("k"), col("v"), col("nv"), col("a")
    kw = dict(select=[k, sum_(v), sum_(v * 1.5 - 2.0), sum_(If(a > 30000, nv, 0.0)), count(1)],
              group_by=[k], where=W)
    plan = Plan(T.MIXED_SCHEMA, float_sum_mode=K.FLOAT_SUM_EXACT, **kw)
    exp = O.oracle_run(img, Plan(T.MIXED_SCHEMA, **kw))
    q = t.query(plan)
    r1 = q.run().rows()
    r2 = q.run().rows()
    q.close()
    assert sorted(map(repr, r1)) == sorted(map(repr, r2))          # run to run
    T.compare_results(r1, exp.rows(), exp.types, rel=1e-12)
    # the exactly rounded sum of the passing rows of every group
    passing = (c["a"] > 30000) & (c["b"] < 30000)
    want = {}
    for kk in np.unique(c["k"][passing]):
        want[int(kk)] = math.fsum(c["v"][passing & (c["k"] == kk)].tolist())
    got = {r[0]: r[1] for r in r1}
    assert got == want
    # any split of the rows over partitions, same bound everywhere: the same bits
    bound = 1e6
    single = t.query(Plan(T.MIXED_SCHEMA, float_sum_mode=K.FLOAT_SUM_EXACT, float_sum_bound=bound, **kw))
    base = sorted(map(repr, single.run().rows()))
    single.close()
    for cuts in ([0, 100_000, 300_000], [0, 1, 299_999, 300_000], [0, 77_777, 155_555, 300_000]):
        nr = len(cuts) - 1
        hub = E.Hub(nr)
        out = [None] * nr

        def work(r):
            cx = E.Context(0)
            tt = cx.open_image(img)
            qq = tt.query(Plan(T.MIXED_SCHEMA, float_sum_mode=K.FLOAT_SUM_EXACT,
                               float_sum_bound=bound, row_begin=cuts[r], row_end=cuts[r + 1], **kw))
            x = E.Exchange.hub(cx, hub, r)
            qq.execute()
            qq.exchange(x, K.EXCHANGE_GATHER_ALL)
            out[r] = sorted(map(repr, qq.fetch_all().rows()))
            qq.close(); x.close(); tt.close(); cx.close()

        th = [threading.Thread(target=work, args=(r,)) for r in range(nr)]
        [x.start() for x in th]
        [x.join(timeout=300) for x in th]
        hub.close()
        assert all(o == base for o in out), cuts
    # a bound the data exceeds, and an argument without a derivable bound
    q = t.query(Plan(T.MIXED_SCHEMA, select=[sum_(v)], float_sum_mode=K.FLOAT_SUM_EXACT,
                     float_sum_bound=10.0))
    with pytest.raises(E.EvqlError) as ei:
        q.run()
    assert ei.value.code == K.EVQL_ERUNTIME and "bound" in ei.value.msg
    q.close()
    with pytest.raises(E.EvqlError) as ei:
        t.query(Plan(T.MIXED_SCHEMA, select=[sum_(v / (nv + 1.0))], float_sum_mode=K.FLOAT_SUM_EXACT))
    assert ei.value.code == K.EVQL_ENOTSUP
    # partial-aggregate rows carry the rounded double (the reference's wire format)
    q = t.query(Plan(T.MIXED_SCHEMA, select=[k, sum_(v)], group_by=[k], mode=K.MODE_PARTIAL,
                     float_sum_mode=K.FLOAT_SUM_EXACT))
    rows = dict(q.run().rows())
    q.close()
    full = {int(kk): math.fsum(c["v"][c["k"] == kk].tolist()) for kk in np.unique(c["k"])}
    import struct
    assert len(rows) == len(full)
    for key, data in list(rows.items())[:50]:
        # data = SValue::encode(k) (1 + 1 + 9 bytes) then the 8 raw bytes of the double
        kk = struct.unpack_from("<Q", data, 2)[0]
        assert struct.unpack_from("<d", data, 11)[0] == full[kk]
