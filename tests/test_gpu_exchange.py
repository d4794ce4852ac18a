"""evql_query_exchange: the exchange step of a GROUP BY over several partitions, behind
the C ABI.  Ranks are threads of this process, every rank with its own context (stream)
on the one GPU of the box, joined by the in-process hub transport; the kernels --
owner bucketing, first-row resolution, string heaps, rank-ordered merges -- are the ones
the RCCL transport drives across GPUs.  The merged result must equal the oracle on the
concatenation of the partitions (the reference: PartialGroupBy per partition +
GroupByMergeExpression, groupby.cc:231-714)."""
import threading

import numpy as np
import pytest

import eventql_amd as E
from eventql_amd import capi as K
from eventql_amd.plan import Agg, Plan, col, count, sum_, min_, max_, mean
import oracle_lib as O
import tables as T

pytestmark = pytest.mark.gpu

COLS = [dict(name="k", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_PLAIN),
        dict(name="a", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT32_BITPACKED),
        dict(name="v", logical_type=K.COL_FLOAT, storage_type=K.ENC_FLOAT_IEEE754),
        dict(name="s", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN),
        dict(name="ns", logical_type=K.COL_STRING, storage_type=K.ENC_STRING_PLAIN, dlevel_max=1),
        dict(name="u", logical_type=K.COL_UNSIGNED_INT, storage_type=K.ENC_UINT64_LEB128)]
S = dict(k=K.T_UINT64, a=K.T_UINT64, v=K.T_FLOAT64, s=K.T_STRING, ns=K.T_STRING, u=K.T_UINT64)


def partition(seed, n):
    rng = np.random.default_rng(seed)
    c = dict(k=rng.integers(0, 500, n, dtype=np.uint64),
             a=rng.integers(0, 65536, n, dtype=np.uint64),
             v=rng.random(n) * 1000.0,
             u=rng.integers(0, 200_000, n, dtype=np.uint64))
    c["s"] = [b"key-%d" % (x % 3000) for x in c["u"]]
    c["ns"] = [b"n%d" % (x % 700) for x in c["a"]]
    c["ns_present"] = (rng.random(n) < 0.8).astype(np.uint8)
    return c


def image_of(parts):
    w = E.Writer(COLS)
    n = sum(len(p["k"]) for p in parts)
    for name in ("k", "a", "v", "u"):
        w.put(name, np.concatenate([p[name] for p in parts]))
    w.put("s", [x for p in parts for x in p["s"]])
    w.put("ns", [x for p in parts for x in p["ns"]],
          present=np.concatenate([p["ns_present"] for p in parts]))
    w.commit(n)
    img = w.image()
    w.close()
    return img


def run_ranks(nranks, parts, plan_kw, mode):
    """-> list (per rank) of result rows"""
    hub = E.Hub(nranks)
    out = [None] * nranks
    errs = []

    def work(r):
        try:
            ctx = E.Context(0)
            t = ctx.open_image(image_of([parts[r]]))
            q = t.query(Plan(S, **plan_kw))
            x = E.Exchange.hub(ctx, hub, r)
            assert x.backend() == "hub"
            q.execute()
            q.exchange(x, mode)
            res = q.fetch_all()
            out[r] = (res.rows(), res.types, x.stats())
            q.close()
            x.close()
            t.close()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            errs.append(e)
            # keep the other ranks from waiting for ever at the hub's barrier
            raise

    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in th), "a rank hangs"
    assert not errs, errs
    hub.close()
    return out


PLANS = {
    "u64-key": dict(select=[col("k"), count(1), sum_(col("a")), sum_(col("v")), min_(col("v")),
                            max_(col("a")), mean(col("v"))], group_by=[col("k")]),
    "string-key": dict(select=[col("s"), count(1), sum_(col("a")), max_(col("v"))],
                       group_by=[col("s")]),
    "nullable-string-key": dict(select=[col("ns"), count(1), sum_(col("a"))], group_by=[col("ns")]),
    "two-keys": dict(select=[col("k"), col("ns"), count(1), sum_(col("u"))],
                     group_by=[col("k"), col("ns")], key_cols=2),
    "high-card-leb-key": dict(select=[col("u"), count(1), sum_(col("a"))], group_by=[col("u")],
                              groups_hint=400_000),
    "string-key-partitioned-path": dict(select=[col("s"), count(1), sum_(col("a"))],
                                        group_by=[col("s")], groups_hint=400_000),
    "where": dict(select=[col("k"), count(1), sum_(col("v"))], group_by=[col("k")],
                  where=(col("a") > 30000) & (col("s") >= "key-2")),
    "global": dict(select=[count(1), sum_(col("a")), max_(col("v"))], group_by=[], key_cols=0),
    # count_distinct: the (group, value) pair sets follow their groups (aggregate.cc:119-137)
    "distinct-u64-key": dict(select=[col("k"), Agg("count_distinct", col("a") % 97), count(1),
                                     Agg("count_distinct", col("u"))], group_by=[col("k")]),
    "distinct-string-key": dict(select=[col("ns"), Agg("count_distinct", col("k")), sum_(col("a"))],
                                group_by=[col("ns")]),
    "distinct-global": dict(select=[Agg("count_distinct", col("k")), count(1)], group_by=[],
                            key_cols=0),
}


@pytest.mark.parametrize("name", sorted(PLANS))
@pytest.mark.parametrize("nranks", [2, 3])
def test_exchange_modes_against_the_oracle(name, nranks):
    kw = dict(PLANS[name])
    kc = kw.pop("key_cols", 1)
    parts = [partition(100 + r, 40_000 + 5000 * r) for r in range(nranks)]
    exp = O.oracle_run(image_of(parts), Plan(S, **kw))
    # non-aggregate first-row values differ legitimately (any partition's first row is
    # "the" first row, groupby.cc:606-610): only keys and aggregates are compared
    # GATHER_ALL: every rank holds the complete result
    res = run_ranks(nranks, parts, kw, K.EXCHANGE_GATHER_ALL)
    for rows, types, st in res:
        assert len(rows) == exp.nrows
        T.compare_results(rows, exp.rows(), exp.types, key_cols=kc, rel=1e-9)
    # ... bit-identical on every rank (merges happen in rank order everywhere)
    canon = [sorted(map(repr, r[0])) for r in res]
    assert all(c == canon[0] for c in canon[1:])
    # BY_OWNER: disjoint key ranges whose union is the result
    res = run_ranks(nranks, parts, kw, K.EXCHANGE_BY_OWNER)
    union = [row for rows, _, _ in res for row in rows]
    assert len(union) == exp.nrows
    T.compare_results(union, exp.rows(), exp.types, key_cols=kc, rel=1e-9)
    if exp.nrows > 50:
        assert all(len(rows) > 0 for rows, _, _ in res), "a rank owns nothing"
        assert all(st["bytes_sent"] > 0 for _, _, st in res)


def test_first_row_values_travel_with_the_groups():
    """select of a non-key column: the value comes from the first row of the lowest
    rank that has the group -- the rows of rank 0's partition for every group it holds"""
    parts = [partition(7, 30_000), partition(8, 30_000)]
    kw = dict(select=[col("k"), col("s"), col("a"), count(1)], group_by=[col("k")])
    res = run_ranks(2, parts, kw, K.EXCHANGE_GATHER_ALL)
    exp0 = O.oracle_run(image_of([parts[0]]), Plan(S, **kw))  # rank 0 alone
    first0 = {r[0]: (r[1], r[2]) for r in exp0.rows()}
    both = O.oracle_run(image_of(parts), Plan(S, **kw))
    counts = {r[0]: r[3] for r in both.rows()}
    for rows, _, _ in res:
        assert len(rows) == both.nrows
        for k, s, a, c in rows:
            assert c == counts[k]
            if k in first0:
                assert (s, a) == first0[k]


def test_global_group_first_row_travels_too():
    """no GROUP BY: the one group's non-aggregate values are those of rank 0's first
    passing row; the aggregates cover every rank"""
    parts = [partition(17, 30_000), partition(18, 30_000), partition(19, 20_000)]
    kw = dict(select=[col("s"), col("a"), col("ns"), count(1), sum_(col("a"))], group_by=[],
              where=col("a") > 60000)
    for mode in (K.EXCHANGE_GATHER_ALL, K.EXCHANGE_BY_OWNER):
        res = run_ranks(3, parts, kw, mode)
        exp0 = O.oracle_run(image_of([parts[0]]), Plan(S, **kw)).rows()[0]
        both = O.oracle_run(image_of(parts), Plan(S, **kw)).rows()[0]
        rows = [row for rows, _, _ in res for row in rows]
        assert len(rows) == (3 if mode == K.EXCHANGE_GATHER_ALL else 1)
        for row in rows:
            assert tuple(row[:3]) == tuple(exp0[:3]) and tuple(row[3:]) == tuple(both[3:])


def test_custom_transport_callbacks():
    """a transport supplied through evql_transport_t (here: python callbacks that move
    the device buffers with torch, two ranks in two threads)"""
    import torch
    nranks = 2
    parts = [partition(31, 20_000), partition(32, 20_000)]
    kw = dict(select=[col("s"), count(1), sum_(col("a"))], group_by=[col("s")])
    exp = O.oracle_run(image_of(parts), Plan(S, **kw))
    barrier = threading.Barrier(nranks)
    shared = dict(gather=[None] * nranks, send=[None] * nranks)
    out = [None] * nranks

    def wrap(ptr, nwords):
        # a view of raw device memory as an int64 tensor
        import ctypes as C
        class Holder:  # noqa: E306
            pass
        h = Holder()
        h.__cuda_array_interface__ = dict(shape=(max(int(nwords), 1),), typestr="<i8",
                                          data=(int(ptr), False), version=2)
        return torch.as_tensor(h, device="cuda")[:int(nwords)]

    def work(r):
        ctx = E.Context(0)
        t = ctx.open_image(image_of([parts[r]]))
        q = t.query(Plan(S, **kw))

        def all_gather(send):
            shared["gather"][r] = list(send)
            barrier.wait()
            res = [v for rr in range(nranks) for v in shared["gather"][rr]]
            barrier.wait()
            return res

        def all_to_all(d_send, sc, d_recv, rc, stream):
            torch.cuda.synchronize()
            shared["send"][r] = (d_send, list(sc))
            barrier.wait()
            roff = 0
            for src in range(nranks):
                sp, scs = shared["send"][src]
                soff = sum(scs[:r])
                cnt = scs[r]
                assert cnt == rc[src]
                if cnt:
                    wrap(d_recv + 8 * roff, cnt).copy_(wrap(sp + 8 * soff, cnt))
                roff += cnt
            torch.cuda.synchronize()
            barrier.wait()

        x = E.Exchange.custom(ctx, nranks, r, all_gather, all_to_all, name="python")
        assert x.backend() == "python"
        q.execute()
        q.exchange(x, K.EXCHANGE_BY_OWNER)
        out[r] = q.fetch_all().rows()
        q.close()
        x.close()
        t.close()
        ctx.close()

    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert all(o is not None for o in out)
    union = out[0] + out[1]
    T.compare_results(union, exp.rows(), exp.types, key_cols=1)


def test_rccl_transport_single_rank(ctx):
    """the built-in RCCL transport (ncclCommInitRank, grouped ncclSend / ncclRecv,
    ncclAllGather) with one rank on the one GPU of this box: everything is sent to
    itself, which drives the same calls as N ranks over xGMI"""
    parts = [partition(55, 50_000)]
    img = image_of(parts)
    t = ctx.open_image(img)
    x = E.Exchange.rccl(ctx, 1, 0, E.Exchange.rccl_unique_id())
    assert x.backend() == "rccl"
    try:
        for name in ("u64-key", "string-key", "two-keys", "high-card-leb-key"):
            kw = dict(PLANS[name])
            kc = kw.pop("key_cols", 1)
            exp = O.oracle_run(img, Plan(S, **kw))
            for mode in (K.EXCHANGE_GATHER_ALL, K.EXCHANGE_BY_OWNER):
                q = t.query(Plan(S, **kw))
                q.execute()
                q.exchange(x, mode)
                res = q.fetch_all()
                assert res.nrows == exp.nrows
                T.compare_results(res.rows(), exp.rows(), exp.types, key_cols=kc, rel=1e-9)
                # a second execute + exchange of the same operator (bench.py's step)
                q.execute()
                q.exchange(x, mode)
                assert q.fetch_all().nrows == exp.nrows
                q.close()
    finally:
        x.close()
        t.close()


@pytest.mark.parametrize("name", ["u64-key", "string-key-partitioned-path", "distinct-string-key",
                                  "two-keys", "high-card-leb-key"])
def test_eight_ranks_on_one_gpu(name):
    """the rank count of the target node (8 x MI355X): eight hub ranks as threads on the
    one GPU of this box, both modes -- config 3's shape (u64 key, float sums), config 4's
    (high cardinality, the partitioned path), config 4s' (string keys: first rows and string
    heaps travel), count_distinct pair sets, two-column keys"""
    kw = dict(PLANS[name])
    kc = kw.pop("key_cols", 1)
    parts = [partition(900 + r, 12_000 + 1000 * r) for r in range(8)]
    exp = O.oracle_run(image_of(parts), Plan(S, **kw))
    res = run_ranks(8, parts, kw, K.EXCHANGE_BY_OWNER)
    union = [row for rows, _, _ in res for row in rows]
    assert len(union) == exp.nrows
    T.compare_results(union, exp.rows(), exp.types, key_cols=kc, rel=1e-9)
    assert sum(1 for rows, _, _ in res if rows) == 8, "a rank owns nothing"
    res = run_ranks(8, parts, kw, K.EXCHANGE_GATHER_ALL)
    canon = [sorted(map(repr, r[0])) for r in res]
    assert all(c == canon[0] for c in canon[1:])
    T.compare_results(res[7][0], exp.rows(), exp.types, key_cols=kc, rel=1e-9)


def test_eight_ranks_exact_float_sums():
    """EVQL_FLOAT_SUM_EXACT with one bound on every rank: the merged sums do not depend on
    how the rows are split over 8 ranks (bit-identical to the 1-rank result)"""
    parts = [partition(950 + r, 9_000) for r in range(8)]
    kw = dict(select=[col("k"), sum_(col("v")), count(1)], group_by=[col("k")],
              float_sum_mode=K.FLOAT_SUM_EXACT, float_sum_bound=1024.0)
    one = run_ranks(1, [dict((n, np.concatenate([p[n] for p in parts]) if n not in ("s", "ns")
                              else [x for p in parts for x in p[n]]) for n in parts[0])],
                    kw, K.EXCHANGE_GATHER_ALL)
    eight = run_ranks(8, parts, kw, K.EXCHANGE_GATHER_ALL)
    base = sorted(map(repr, one[0][0]))
    assert all(sorted(map(repr, r[0])) == base for r in eight)


def test_wide_first_row_plan_is_refused_before_anything_is_written(ctx):
    """the merged slot of a plan that reads first-row values is W + ncols + 1
    words; beyond the 33 the merge kernels take, the exchange answers EVQL_ENOTSUP -- before
    it fills any fixed-size array"""
    parts = [partition(77, 5_000)]
    t = ctx.open_image(image_of(parts))
    aggs = [min_(col("a") + i) for i in range(13)]           # 26 state words
    kw = dict(select=[col("k"), col("s"), col("ns"), col("u"), col("v"), col("a")] + aggs,
              group_by=[col("k")])
    q = t.query(Plan(S, **kw))
    hub = E.Hub(1)
    x = E.Exchange.hub(ctx, hub, 0)
    q.execute()
    with pytest.raises(E.EvqlError) as ei:
        q.exchange(x, K.EXCHANGE_GATHER_ALL)
    assert ei.value.code == K.EVQL_ENOTSUP and "words per merged group" in ei.value.msg
    # the un-exchanged result is still there
    assert q.fetch_all().nrows == len(set(parts[0]["k"].tolist()))
    q.close()
    x.close()
    hub.close()
    t.close()


@pytest.mark.parametrize("name", ["distinct-u64-key", "distinct-string-key", "distinct-global"])
def test_partial_rows_with_count_distinct_after_an_exchange(name):
    """EVQL_MODE_PARTIAL + count_distinct through the device exchange: the pair sets follow
    their groups, and the PartialGroupByExpression rows emitted afterwards carry the MERGED
    sets' values (varuint size + ascending values, aggregate.cc:111-117) -- byte for byte
    what the oracle's partial operator emits over the concatenated partitions"""
    kw = dict(PLANS[name])
    kw.pop("key_cols", None)
    parts = [partition(300 + r, 20_000 + 3000 * r) for r in range(3)]
    pkw = dict(kw, mode=K.MODE_PARTIAL)
    ep = O.oracle_run(image_of(parts), Plan(S, **pkw))
    want = sorted((ep.keys[20 * i:20 * i + 20], ep.columns[0][i]) for i in range(ep.nrows))
    res = run_ranks(3, parts, pkw, K.EXCHANGE_GATHER_ALL)
    for rows, _, _ in res:
        assert sorted(rows) == want
    res = run_ranks(3, parts, pkw, K.EXCHANGE_BY_OWNER)
    assert sorted(r for rows, _, _ in res for r in rows) == want


def big_partition(seed, n, spread):
    c = partition(seed, n)
    rng = np.random.default_rng(seed + 77)
    c["u"] = rng.integers(0, spread, n, dtype=np.uint64)
    c["s"] = [b"key-%d" % x for x in c["u"]]
    return c


BIG_PLANS = {
    # config 4's shape: exact u64 identities, integer and float states
    "u64-key": dict(select=[col("u"), count(1), sum_(col("a")), sum_(col("v")), min_(col("v")),
                            max_(col("a"))], group_by=[col("u")], groups_hint=500_000),
    # config 4s': string keys -- first rows resolved by the sender, bytes in the heap
    "string-key": dict(select=[col("s"), count(1), sum_(col("a"))], group_by=[col("s")],
                       groups_hint=500_000),
    # two key columns: 128-bit hashed identities, key values from the first row
    "two-keys": dict(select=[col("u"), col("k"), count(1), max_(col("a"))],
                     group_by=[col("u"), col("k")], key_cols=2, groups_hint=500_000),
    "exact-sums": dict(select=[col("u"), sum_(col("v")), count(1)], group_by=[col("u")],
                       groups_hint=500_000, float_sum_mode=K.FLOAT_SUM_EXACT, float_sum_bound=1024.0),
}


@pytest.mark.parametrize("name", sorted(BIG_PLANS))
def test_large_record_sets_are_merged_bucket_by_bucket(name):
    """>= 2^18 received records: split by identity hash into LDS-sized buckets and merged
    there (k_bucket_scatter x 2, k_bucket_merge) instead of through an HBM table; the
    result is the same set of groups"""
    kw = dict(BIG_PLANS[name])
    kc = kw.pop("key_cols", 1)
    nranks = 3
    parts = [big_partition(300 + r, 340_000 + 10_000 * r, 1_500_000) for r in range(nranks)]
    exp = O.oracle_run(image_of(parts), Plan(S, **kw))
    res = run_ranks(nranks, parts, kw, K.EXCHANGE_BY_OWNER)
    assert all(st["merge_buckets"] > 0 for _, _, st in res), [st for _, _, st in res]
    union = [row for rows, _, _ in res for row in rows]
    assert len(union) == exp.nrows
    T.compare_results(union, exp.rows(), exp.types, key_cols=kc, rel=1e-9)
    res = run_ranks(nranks, parts, kw, K.EXCHANGE_GATHER_ALL)
    assert all(st["merge_buckets"] > 0 for _, _, st in res)
    for rows, _, _ in res[:2]:
        assert len(rows) == exp.nrows
        T.compare_results(rows, exp.rows(), exp.types, key_cols=kc, rel=1e-9)
    if name == "exact-sums":
        canon = [sorted(map(repr, r[0])) for r in res]
        assert all(c == canon[0] for c in canon[1:])


def test_first_rows_survive_a_bucketed_merge():
    """the first-row values of a group come from the lowest rank that has it, also when the
    records are merged in the LDS (the record with the smallest (rank << 44 | row) word)"""
    parts = [big_partition(400 + r, 300_000, 600_000) for r in range(2)]
    kw = dict(select=[col("u"), col("s"), col("a"), count(1)], group_by=[col("u")],
              groups_hint=400_000)
    res = run_ranks(2, parts, kw, K.EXCHANGE_GATHER_ALL)
    assert all(st["merge_buckets"] > 0 for _, _, st in res)
    exp0 = O.oracle_run(image_of([parts[0]]), Plan(S, **kw))
    exp1 = O.oracle_run(image_of([parts[1]]), Plan(S, **kw))
    first = {r[0]: (r[1], r[2]) for r in exp1.rows()}
    first.update({r[0]: (r[1], r[2]) for r in exp0.rows()})
    both = O.oracle_run(image_of(parts), Plan(S, **kw))
    counts = {r[0]: r[3] for r in both.rows()}
    rows = res[1][0]
    assert len(rows) == both.nrows
    for u, s, a, c in rows:
        assert c == counts[u] and (s, a) == first[u]


def test_partial_rows_after_a_bucketed_merge():
    """EVQL_MODE_PARTIAL plans exchange their groups the same way: the rows emitted from the
    dense merged records are the PartialGroupByExpression rows of the concatenated table"""
    parts = [big_partition(500 + r, 330_000, 1_200_000) for r in range(3)]
    pkw = dict(select=[col("s"), count(1), sum_(col("a"))], group_by=[col("s")],
               groups_hint=500_000, mode=K.MODE_PARTIAL)
    ep = O.oracle_run(image_of(parts), Plan(S, **pkw))
    want = sorted((ep.keys[20 * i:20 * i + 20], ep.columns[0][i]) for i in range(ep.nrows))
    res = run_ranks(3, parts, pkw, K.EXCHANGE_BY_OWNER)
    assert all(st["merge_buckets"] > 0 for _, _, st in res)
    assert sorted(r for rows, _, _ in res for r in rows) == want
