"""The N > 1 flow of bench.py with the HIP kernels: two ranks share the one GPU of
the test box and exchange their partial aggregates through torch.distributed
(`gloo`, device records staged through the host -- RCCL needs one GPU per rank).
Both exchange shapes: all_gather of dense records (low cardinality) and the
hash-partitioned all_to_all (high cardinality); results vs the oracle."""
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
    return T.MIXED_SCHEMA, Plan, low, high


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        import eventql_amd as E
        from eventql_amd import distributed as D
        import tables as T
        dist.init_process_group("gloo", rank=rank, world_size=world)
        S, Plan, low, high = _plans()
        img, _ = T.mixed_table(N_ROWS)
        ctx = E.Context(0)
        t = ctx.open_image(img)
        cut = [0, 123_457, N_ROWS]
        rng = dict(row_begin=cut[rank], row_end=cut[rank + 1])

        # low cardinality: all_gather, every rank ends with the full result
        ql = t.query(Plan(S, **low, **rng))
        ql.launch()
        ql.finish()
        rw = ql.record_words()
        send = torch.zeros(4096 * rw, dtype=torch.int64, device="cuda")
        n = ql.export_groups(send.data_ptr(), 4096)
        parts = D.exchange_all_gather(send, n, rw, 4096)
        foreign, cnt = D.gather_foreign(parts, rank, rw)
        if cnt:
            ql.import_groups(foreign.data_ptr(), cnt)
        low_rows = ql.fetch_all().rows()

        # high cardinality: all_to_all by identity % world, results stay distributed
        qh = t.query(Plan(S, **high, **rng))
        qm = t.query(Plan(S, **high))
        qh.launch()
        qh.finish()
        rwh = qh.record_words()
        sendh = torch.zeros(400_016 * rwh, dtype=torch.int64, device="cuda")
        nh = qh.export_groups(sendh.data_ptr(), 400_016)
        recv, cnth = D.exchange_all_to_all(sendh, nh, rwh)
        qm.reset()
        if cnth:
            qm.import_groups(recv.data_ptr(), cnth)
        high_rows = qm.fetch_all().rows()
        q.put((rank, low_rows, high_rows))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "ERR " + traceback.format_exc(), None))
        raise


def test_two_ranks_on_one_gpu(built):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import tables as T
    S, Plan, low, high = _plans()
    img, _ = T.mixed_table(N_ROWS)
    exp_low = O.oracle_run(img, Plan(S, **low))
    exp_high = O.oracle_run(img, Plan(S, **high))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(2):
        r, lo, hi = q.get(timeout=600)
        assert not (isinstance(lo, str) and lo.startswith("ERR")), lo
        results[r] = (lo, hi)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # every rank holds the complete low-cardinality result
    for r in range(2):
        T.compare_results(results[r][0], exp_low.rows(), exp_low.types)
    # the high-cardinality result is split by key ownership: disjoint, complete
    k0 = {row[0] for row in results[0][1]}
    k1 = {row[0] for row in results[1][1]}
    assert not (k0 & k1)
    assert all(k % 2 == 0 for k in k0) and all(k % 2 == 1 for k in k1)
    T.compare_results(results[0][1] + results[1][1], exp_high.rows(), exp_high.types)
