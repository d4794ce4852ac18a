"""The N > 1 flow of bench.py across PROCESSES: two ranks share the one GPU of the test
box (RCCL needs one GPU per rank), so the exchange step `evql_query_exchange` runs over
the gloo transport callbacks of eventql_amd.distributed (device words staged through the
host).  Everything else -- owner bucketing, first-row resolution, string heaps, the
rank-ordered merge kernels -- is what the RCCL transport drives.  Both modes, u64 and
string keys; results vs the oracle on the whole table."""
import os
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_ROWS = 300_000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _plans():
    import tables as T
    from eventql_amd.plan import Plan, col, count, sum_, min_, max_
    low = dict(select=[col("k"), count(1), sum_(col("a")), min_(col("b")), max_(col("v")),
                       sum_(col("v"))], group_by=[col("k")], where=col("a") > 5000)
    high = dict(select=[col("w"), count(1), sum_(col("a"))], group_by=[col("w")],
                groups_hint=400_000)
    strk = dict(select=[col("s"), col("ns"), count(1), sum_(col("a"))],
                group_by=[col("s"), col("ns")])
    return T.MIXED_SCHEMA, Plan, low, high, strk


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        import eventql_amd as E
        from eventql_amd import capi as K, distributed as D
        import tables as T
        dist.init_process_group("gloo", rank=rank, world_size=world)
        S, Plan, low, high, strk = _plans()
        img, _ = T.mixed_table(N_ROWS)
        ctx = E.Context(0)
        t = ctx.open_image(img)
        x = D.make_exchange(ctx)
        assert x.backend().startswith("gloo")
        cut = [0, 123_457, N_ROWS]
        rng = dict(row_begin=cut[rank], row_end=cut[rank + 1])
        out = []
        for kw, mode in ((low, K.EXCHANGE_GATHER_ALL), (high, K.EXCHANGE_BY_OWNER),
                         (strk, K.EXCHANGE_BY_OWNER), (strk, K.EXCHANGE_GATHER_ALL)):
            qq = t.query(Plan(S, **kw, **rng))
            qq.launch()
            qq.finish()
            qq.exchange(x, mode)
            out.append(qq.fetch_all().rows())
            qq.close()
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "ERR " + traceback.format_exc()))
        raise


def test_two_ranks_on_one_gpu(built):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import tables as T
    S, Plan, low, high, strk = _plans()
    img, _ = T.mixed_table(N_ROWS)
    exp_low = O.oracle_run(img, Plan(S, **low))
    exp_high = O.oracle_run(img, Plan(S, **high))
    exp_str = O.oracle_run(img, Plan(S, **strk))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(2):
        r, out = q.get(timeout=600)
        assert not isinstance(out, str), out
        results[r] = out
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # GATHER_ALL: every rank holds the complete result
    for r in range(2):
        T.compare_results(results[r][0], exp_low.rows(), exp_low.types)
        T.compare_results(results[r][3], exp_str.rows(), exp_str.types, key_cols=2)
    assert sorted(map(repr, results[0][0])) == sorted(map(repr, results[1][0]))
    # BY_OWNER: disjoint key ranges, complete union
    k0 = {row[0] for row in results[0][1]}
    k1 = {row[0] for row in results[1][1]}
    assert k0 and k1 and not (k0 & k1)
    T.compare_results(results[0][1] + results[1][1], exp_high.rows(), exp_high.types)
    s0 = {row[:2] for row in results[0][2]}
    s1 = {row[:2] for row in results[1][2]}
    assert s0 and s1 and not (s0 & s1)
    T.compare_results(results[0][2] + results[1][2], exp_str.rows(), exp_str.types, key_cols=2)
