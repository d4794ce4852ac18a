"""GroupByMergeExpression over partial-aggregate wire rows (evql_merge_*; host
only, so these run without a GPU).  Frames come from the oracle's
PartialGroupBy restatement, which is pinned on the reference's own bytes
(SURVEY.md 8a a16 / 8c)."""
import numpy as np
import pytest

import eventql_amd as E
from eventql_amd import capi as K
from eventql_amd.plan import (Plan, Col, Lit, Call, If, count, sum_, min_, max_, mean,
                              count_distinct)
import oracle_lib as O
import tables as T

N = 200_000
S = T.SURVEY_SCHEMA
col = Col


@pytest.fixture(scope="module")
def survey(built):
    return T.survey_table(N)[0]


def _ranges(n, parts):
    step = (n + parts - 1) // parts
    return [(i * step, min(n, (i + 1) * step)) for i in range(parts)]


def _part(lo, hi, mode=K.MODE_PARTIAL, **kw):
    """plan over rows [lo, hi): the partitions are cut with the external row
    filter (FastCSTableScan::setFilter, CSTableScan.cc:1006-1009)"""
    mask = np.zeros(N, np.uint8)
    mask[lo:hi] = 1
    return Plan(S, mode=mode, row_filter=mask, **kw)


def _merge(plan_kw, img, parts=3):
    frames = [O.oracle_partial_frame(img, _part(lo, hi, **plan_kw))
              for lo, hi in _ranges(N, parts)]
    plan = Plan(S, **plan_kw)
    m = E.Merge(plan)
    for f in frames:
        m.add_frame(f)
    got = m.fetch_all()
    exp_merge = O.oracle_merge(plan, frames)
    exp_full = O.oracle_run(img, plan)
    m.close()
    return got, exp_merge, exp_full


def test_reference_verified_bytes(built):
    """SURVEY.md 8a a16 [ran]: `select k, sum(a), count(1) .. group by k`, group
    k=355 on the reference emits key 72d31a72.. and data
    01 09 6301000000000000 00 | 8c89f60e | b607"""
    key = bytes.fromhex("72d31a72564b88de26a4bd48ce436c184f0d38f4")
    data = bytes.fromhex("01096301000000000000008c89f60eb607")
    plan = Plan(S, select=[col("k"), sum_(col("a")), count(1)], group_by=[col("k")])
    m = E.Merge(plan)
    m.add_frame(O.partial_frame([key], [data]))
    assert m.fetch_all().rows() == [(355, 31294604, 950)]
    m.add_frame(O.partial_frame([key], [data], flags=1))
    assert m.num_groups == 1
    assert m.fetch_all().rows() == [(355, 2 * 31294604, 1900)]
    m.close()


@pytest.mark.parametrize("kw,nkeys,full", [
    (dict(select=[col("k"), sum_(col("a")), count(1), sum_(col("b"))], group_by=[col("k")],
          where=(col("a") > 30000) & (col("b") < 30000)), 1, True),
    (dict(select=[col("k"), col("s"), count(1), min_(col("a")), max_(col("v")), mean(col("b")),
                  sum_(col("v"))], group_by=[col("k"), col("s")]), 2, True),
    # NULL key group, min/mean over a nullable column, expression key not selected
    (dict(select=[count(col("n")), sum_(col("n")), min_(col("n")), mean(col("n"))],
          group_by=[col("n") % 7]), 0, True),
    # first-row column: merged value = last frame's (see the next test), not the full scan's
    (dict(select=[col("k"), col("a"), count(1)], group_by=[col("k")]), 1, False),
    (dict(select=[count(1), sum_(col("a")) + count(1), max_(col("b"))]), 0, True),
    # count_distinct states are value sets: the merge unions them
    (dict(select=[col("k"), count_distinct(col("b") % 50), count_distinct(col("n"))],
          group_by=[col("k")]), 1, True),
])
def test_merge_of_row_range_partials(survey, kw, nkeys, full):
    got, exp_merge, exp_full = _merge(kw, survey)
    assert got.types == exp_merge.types == exp_full.types
    assert got.nrows == exp_merge.nrows == exp_full.nrows
    if nkeys:
        # same merge order as the restatement: bit-exact, floats included
        T.compare_results(got.rows(), exp_merge.rows(), got.types, key_cols=nkeys, rel=0)
        if full:
            T.compare_results(got.rows(), exp_full.rows(), got.types, key_cols=nkeys, rel=1e-9)
    else:
        assert sorted(map(repr, got.rows())) == sorted(map(repr, exp_merge.rows()))
        if full:
            assert sorted(map(repr, got.rows())) == sorted(map(repr, exp_full.rows()))


def test_non_aggregates_take_the_last_decoded_value(survey):
    """groupby.cc:606-610: SValue::decode overwrites; first-row columns of the
    merged result therefore come from the LAST frame that carried the group"""
    kw = dict(select=[col("k"), col("b"), count(1)], group_by=[col("k")])
    parts = _ranges(N, 2)
    frames = [O.oracle_partial_frame(survey, _part(lo, hi, **kw)) for lo, hi in parts]
    last = O.oracle_run(survey, _part(parts[1][0], parts[1][1], mode=K.MODE_FINAL, **kw))
    plan = Plan(S, **kw)
    m = E.Merge(plan)
    for f in frames:
        m.add_frame(f)
    got = {r[0]: r for r in m.fetch_all().rows()}
    exp = {r[0]: r for r in O.oracle_merge(plan, frames).rows()}
    assert got == exp
    for k, b, _ in last.rows():
        assert got[k][1] == b
    m.close()


def test_add_rows_takes_next_batch_vectors(survey):
    kw = dict(select=[col("s"), count(1), sum_(col("a"))], group_by=[col("s")])
    plan = Plan(S, **kw)
    m = E.Merge(plan)
    for lo, hi in _ranges(N, 2):
        r = O.oracle_run(survey, _part(lo, hi, **kw))
        keys_raw = b"".join((20).to_bytes(4, "little") + r.keys[20 * i:20 * i + 20] + b"\0"
                            for i in range(r.nrows))
        m.add_rows(keys_raw, r.raw[0], r.nrows)
    exp = O.oracle_run(survey, plan)
    T.compare_results(m.fetch_all().rows(), exp.rows(), exp.types, key_cols=1)
    m.close()


def test_malformed_frames_are_rejected(survey):
    kw = dict(select=[col("k"), sum_(col("a"))], group_by=[col("k")])
    frame = O.oracle_partial_frame(survey, Plan(S, mode=K.MODE_PARTIAL, row_end=5000, **kw))
    m = E.Merge(Plan(S, **kw))
    for bad in (frame[:-1], frame[:30], b"\x00", O.varuint(0) + O.varuint(3)):
        with pytest.raises(E.EvqlError) as ei:
            m.add_frame(bad)
        assert ei.value.code == K.EVQL_EIO
        assert "invalid partialaggr result encoding" in ei.value.msg
    with pytest.raises(RuntimeError):
        O.oracle_merge(Plan(S, **kw), [frame[:-1]])
    m.close()
