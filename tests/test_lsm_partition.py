"""GROUP BY over an evqld partition: eventql::PartitionCursor (server/sql/
partition_cursor.cc:34-235) pinned on the reference's own implementation.

tests/golden/ref_csql_lsm.json holds what the reference engine returned when its
PartitionCursor -- driven by the probe from a hand-made PartitionSnapshot over the LSM
files of tests/lsm_tables.py (oracle/ref_csql/probe.cc PARTITION) -- fed its
GroupByExpression / PartialGroupByExpression, and, as `select rid`, the row filters
themselves (every row the cursor let through).  Against these:

  CPU   oracle/lsm_oracle.c (the filter loop incl. the two "no filter" shortcuts) and
        orc_query_run_chain (one group map over the scans of a chain)
  GPU   evql_lsm_chain_* (filters on the device), evql_query_create_chain (one operator
        over the partition) and the reference engine with the GPU operator plugged in,
        finding the partition through the adapter's resolver (gpu_partition.h)
"""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

import eventql_amd as E
from eventql_amd import capi as K
from eventql_amd.plan import Plan, DumpedPlan, dump_of_plan
import lsm_tables
import oracle_lib as O
import refcases
import sqlgen
import tables as T
from test_ref_csql_cpu import check_result, check_partial, all_lowerable, strip_dump

PROBE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                     "oracle", "_ref", "csql_probe")
FIXTURE = os.path.join(T.GOLDEN, "ref_csql_lsm.json")


def load():
    return json.load(open(FIXTURE))


_fx = {}
_cases = {}


def fixture_case(cid):
    if not _fx:
        _fx.update({c["id"]: c for c in load()["cases"]})
    return _fx[cid]


def case_plan(cid):
    if not _cases:
        _cases.update({c["id"]: c for c in refcases.SUITES["lsm"]()})
    return _cases[cid]


def _ids():
    return [c["id"] for c in load()["cases"]]


def make_plan(c, **extra):
    return Plan(lsm_tables.LSM_SCHEMA, scan_mode=c["scan_mode"], **dict(c["kw"], **extra))


_oracle_filters = {}


def oracle_filters(pname):
    if pname not in _oracle_filters:
        _oracle_filters[pname] = O.oracle_partition_filters(lsm_tables.partition(pname))
    return _oracle_filters[pname]


def scan_order_images(pname):
    return [f[1] for f in reversed(lsm_tables.partition(pname))]


# ---------------------------------------------------------------------------------------
# CPU: the oracle against the reference's PartitionCursor
# ---------------------------------------------------------------------------------------
def test_fixture_matches_case_list():
    fx = load()
    cases = refcases.SUITES["lsm"]()
    assert [c["id"] for c in fx["cases"]] == sorted(c["id"] for c in cases)
    by_id = {c["id"]: c for c in cases}
    for c in fx["cases"]:
        assert c["sql"] == by_id[c["id"]]["sql"], c["id"]


@pytest.mark.parametrize("pname", sorted(lsm_tables.PARTITIONS))
def test_filter_restatement_agrees_with_the_python_model(pname):
    """two independent restatements of partition_cursor.cc:134-195 (C and python); the
    reference itself is compared in test_oracle_reproduces_the_partition_cursor"""
    files = lsm_tables.partition(pname)
    got = oracle_filters(pname)
    exp = lsm_tables.model_filters(files)
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert (g is None) == (e is None)
        if g is not None:
            assert (g == e).all()


def test_plan_py_emits_the_reference_compilers_bytecode():
    n = 0
    for fx in load()["cases"]:
        if "programs" not in fx or not all_lowerable(fx["programs"]):
            continue
        assert fx["programs"]["symbols_agree"], fx["id"]
        c = case_plan(fx["id"])
        if "select" not in c["kw"]:
            continue
        assert dump_of_plan(make_plan(c)) == strip_dump(fx["programs"]), (fx["id"], fx["sql"])
        n += 1
    assert n >= 50


@pytest.mark.parametrize("cid", _ids())
def test_oracle_reproduces_the_partition_cursor(cid):
    fx = fixture_case(cid)
    c = case_plan(cid)
    pname = c["table"][4:]
    imgs = scan_order_images(pname)
    filters = oracle_filters(pname)
    try:
        r = O.oracle_run_chain(imgs, filters, make_plan(c))
        got = (r.types, r.rows())
    except RuntimeError as e:
        got = str(e)
    check_result(fx["result"], got)
    if cid.endswith("-scan"):
        # the filters themselves: the rids the reference's cursor let through
        kept = []
        for f, (_, _, _, _, cols) in zip(filters, reversed(lsm_tables.partition(pname))):
            kept += [int(x) for x in (cols["rid"] if f is None else cols["rid"][f])]
        assert sorted(kept) == sorted(int(x[0]) for x in got[1])
        return
    if "programs" in fx and all_lowerable(fx["programs"]):
        r2 = O.oracle_run_chain(imgs, filters, DumpedPlan(fx["programs"], scan_mode=c["scan_mode"]))
        check_result(fx["result"], (r2.types, r2.rows()))
    if fx["partial"]["ok"]:
        rp = O.oracle_run_chain(imgs, filters, make_plan(c, mode=K.MODE_PARTIAL))
        keys = [rp.keys[20 * i:20 * i + 20] for i in range(rp.nrows)]
        check_partial(fx["partial"], keys, rp.columns[0])


# ---------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------
_chains = {}


@pytest.fixture(scope="module")
def gpu_chain(ctx):
    def get(pname):
        if pname not in _chains:
            files = lsm_tables.partition(pname)
            tabs = [ctx.open_image(f[1]) for f in reversed(files)]
            ch = E.LsmChain(ctx)
            for t, f in zip(tabs, reversed(files)):
                ch.add(t, has_skiplist=f[2], has_updates=f[3])
            ch.build()
            _chains[pname] = (ch, tabs)
        return _chains[pname][0]
    yield get
    for ch, tabs in _chains.values():
        ch.close()
        for t in tabs:
            t.close()
    _chains.clear()


@pytest.mark.gpu
@pytest.mark.parametrize("pname", sorted(lsm_tables.PARTITIONS))
def test_device_filters_are_the_partition_cursors(gpu_chain, pname):
    ch = gpu_chain(pname)
    exp = oracle_filters(pname)
    files = list(reversed(lsm_tables.partition(pname)))
    for i, e in enumerate(exp):
        f, kept = ch.filter(i)
        if e is None:
            assert f is None, (pname, i)          # setFilter is not called
            assert kept == len(files[i][4]["ids"])
        else:
            assert f is not None and (f == e).all(), (pname, i)
            assert kept == int(e.sum())


def run_chain_query(ch, plan):
    try:
        q = ch.query(plan)
    except E.EvqlError as e:
        if e.code == K.EVQL_ENOTSUP:
            return None
        raise
    try:
        try:
            r = q.run()
        except E.EvqlError as e:
            return e.msg
        return (r.types, r.rows())
    finally:
        q.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cid", [i for i in _ids() if not i.endswith("-scan")])
def test_hip_chain_operator_reproduces_the_reference(gpu_chain, cid):
    fx = fixture_case(cid)
    c = case_plan(cid)
    ch = gpu_chain(c["table"][4:])
    r = run_chain_query(ch, make_plan(c))
    assert r is not None, "not lowered"
    check_result(fx["result"], r)
    if "programs" in fx and all_lowerable(fx["programs"]):
        r2 = run_chain_query(ch, DumpedPlan(fx["programs"], scan_mode=c["scan_mode"]))
        assert r2 is not None
        check_result(fx["result"], r2)
    if fx["partial"]["ok"] and fx["result"]["ok"]:
        rp = run_chain_query(ch, make_plan(c, mode=K.MODE_PARTIAL))
        assert rp is not None and not isinstance(rp, str), rp
        rows = rp[1]
        check_partial(fx["partial"], [k for k, _ in rows], [d for _, d in rows])


@pytest.mark.gpu
def test_chain_operator_extras(gpu_chain, ctx):
    """what the fixtures cannot say: ORDER BY .. LIMIT over the merged groups, float
    sums / min / max across the chain (build-supplied aggregates: against the oracle),
    a second execute, and the refusals"""
    from eventql_amd.plan import col, count, sum_, min_, max_, Order
    ch = gpu_chain("basic")
    imgs = scan_order_images("basic")
    filters = oracle_filters("basic")
    S = lsm_tables.LSM_SCHEMA
    kw = dict(select=[col("k"), count(1), sum_(col("v")), min_(col("a")), max_(col("n")), col("s")],
              group_by=[col("k")])
    exp = O.oracle_run_chain(imgs, filters, Plan(S, **kw))
    q = ch.query(Plan(S, **kw))
    for _ in range(2):  # executes twice: the merged table is rebuilt
        q.execute()
        got = q.fetch_all()
        T.compare_results(got.rows(), exp.rows(), exp.types)
    st = q.stats()
    assert st["rows_scanned"] == sum(len(f[4]["ids"]) for f in lsm_tables.partition("basic"))
    assert st["rows_passed"] == sum(len(f[4]["ids"]) if x is None else int(x.sum())
                                 for f, x in zip(reversed(lsm_tables.partition("basic")), filters))
    # a merged result cannot be exported / imported / viewed as a partial table
    import torch
    buf = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
    for fn in (lambda: q.export_groups(buf.data_ptr(), 64), lambda: q.import_groups(buf.data_ptr(), 1)):
        with pytest.raises(E.EvqlError) as ei:
            fn()
        assert ei.value.code in (K.EVQL_EARG, K.EVQL_ENOTSUP)
    q.close()
    # ORDER BY count desc, k limit 5 offset 2 over the chain's groups
    kw2 = dict(select=[col("k"), count(1), sum_(col("a"))], group_by=[col("k")])
    p2 = Plan(S, **kw2)
    order = Order(p2, [(1, True), (0, False)], limit=5, offset=2)
    exp2 = O.oracle_run_chain(imgs, filters, p2)
    rows = sorted(exp2.rows(), key=lambda r: (-r[1], r[0]))[2:7]
    q2 = ch.query(p2)
    q2.set_order(order)
    assert q2.run().rows() == rows
    q2.close()
    # exact float sums: one quantum across the chain, bit-identical from run to run
    kw3 = dict(select=[col("k"), sum_(col("v"))], group_by=[col("k")])
    res = []
    for _ in range(2):
        q3 = ch.query(Plan(S, float_sum_mode=K.FLOAT_SUM_EXACT, **kw3))
        res.append(sorted(q3.run().rows()))
        q3.close()
    assert res[0] == res[1]
    exp3 = {r[0]: r[1] for r in O.oracle_run_chain(imgs, filters, Plan(S, **kw3)).rows()}
    for k, v in res[0]:
        assert abs(v - exp3[k]) <= 1e-9 * abs(exp3[k])
    # row filters / ranges belong to the chain
    with pytest.raises(E.EvqlError) as ei:
        ch.query(Plan(S, row_end=10, **kw2))
    assert ei.value.code == K.EVQL_EARG


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(PROBE), reason="oracle/_ref/csql_probe not built "
                    "(needs /root/reference at build time)")
def test_reference_engine_with_gpu_operator_over_partitions():
    """the reference's own engine, `MODE gpu`: GpuScheduler asks the registry for the
    scan's table, the registry's resolver turns the PartitionSnapshot into the file chain
    (gpu_partition.h), the operator is evql_query_create_chain -- same rows as the
    reference's PartitionCursor + GroupByExpression gave, and as PartialGroupByExpression
    gave in partial mode"""
    fx = load()
    by_table = {}
    for c in fx["cases"]:
        by_table.setdefault(c["table"], []).append(c)
    lowered = groupbys = 0
    with tempfile.TemporaryDirectory() as tmp:
        for table, cs in sorted(by_table.items()):
            specs = []
            for fname, img, skl, upd, _ in refcases.partition_files(table):
                with open(os.path.join(tmp, fname + ".cst"), "wb") as f:
                    f.write(img)
                specs.append("%s:%d:%d" % (fname, skl, upd))
            gb = [c for c in cs if not c["id"].endswith("-scan")]
            cmds = ["PARTITION t %s %s" % (tmp, " ".join(specs)), "ROWS on", "MODE gpu"]
            cmds += ["SQL " + c["sql"] for c in cs]
            cmds += ["MODE gpu partial"] + ["SQL " + c["sql"] for c in gb]
            p = subprocess.run([PROBE], input="\n".join(cmds) + "\n", capture_output=True,
                               text=True, timeout=900)
            assert p.returncode == 0, p.stderr[-2000:]
            res = [json.loads(l) for l in p.stdout.splitlines() if l.strip()]
            assert len(res) == len(cs) + len(gb)
            for c, r in zip(cs, res):
                want = c["result"]
                assert r["ok"], (c["id"], r.get("error"))
                assert r["types"] == want["types"], c["id"]
                rows = [list(x) for x in r["rows"]]
                rows.sort(key=lambda row: [(0, "") if v is None else (1, repr(v)) for v in row])
                assert len(rows) == want["nrows"], c["id"]
                assert rows[:len(want["rows"])] == want["rows"], c["id"]
                assert sqlgen.rows_digest(rows) == want["digest"], c["id"]
                if not c["id"].endswith("-scan"):
                    groupbys += 1
                    d = [x for x in r.get("decisions", []) if x["node"] == "groupby"]
                    lowered += 1 if d and d[0]["lowered"] else 0
            for c, r in zip(gb, res[len(cs):]):
                want = c["partial"]
                if not want["ok"]:
                    continue
                assert r["ok"], (c["id"], r.get("error"))
                pairs = sorted([x[0].encode("latin-1").hex(), x[1].encode("latin-1").hex()]
                               for x in r["rows"])
                assert len(pairs) == want["nrows"], c["id"]
                assert sqlgen.rows_digest(pairs) == want["digest"], c["id"]
    assert lowered >= 0.95 * groupbys, (lowered, groupbys)
