"""world_size-2 `gloo` test of the N>1 path: partition assignment, the dense
group-record wire format and both exchange patterns (all_gather for low
cardinality, hash-partitioned all-to-all for high cardinality).

No GPU here, so every rank's *partial aggregate* comes from the oracle run on
its own partition (standing in for the scan kernel) and the merge of received
records is a numpy restatement of `k_table_merge`; the result must equal the
oracle on the concatenated table.  On the GPU box the same exchange functions
move records produced by `evql_query_export_groups` (tests/test_gpu_parity.py,
bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _partition_columns(part, n):
    sys.path.insert(0, ROOT)
    from eventql_amd import synth
    c = synth.table_columns(n, seed=synth.SEED + 977 * part)
    c["u"] = c["x"] % np.uint64(5000)
    return c


def _write(columns_list, names=("k", "a", "v", "u")):
    import eventql_amd as E
    from eventql_amd import capi as K
    w = E.Writer([
        dict(name=n, logical_type=K.COL_FLOAT if n == "v" else K.COL_UNSIGNED_INT,
             storage_type=K.ENC_FLOAT_IEEE754 if n == "v" else K.ENC_UINT64_PLAIN)
        for n in names])
    total = 0
    for n in names:
        w.put(n, np.concatenate([c[n] for c in columns_list]))
    total = sum(len(c["k"]) for c in columns_list)
    w.commit(total)
    img = w.image()
    w.close()
    return img


def _plan(key):
    from eventql_amd import capi as K
    from eventql_amd.plan import Plan, col, count, sum_, max_
    S = dict(k=K.T_UINT64, a=K.T_UINT64, v=K.T_FLOAT64, u=K.T_UINT64)
    return Plan(S, select=[col(key), count(1), sum_(col("a")), max_(col("a")), sum_(col("v"))],
                group_by=[col(key)], where=col("a") > 20000)


# record layout for that plan: [kind, ident, count, sum_a, max_a, cnt_a, sum_v]
RW = 7


def _records_from_oracle(res):
    rec = np.zeros((res.nrows, RW), dtype=np.int64)
    for i, (k, cnt, sa, ma, sv) in enumerate(res.rows()):
        rec[i, 1] = np.int64(np.uint64(k).view(np.int64)) if False else int(k)
        rec[i, 2] = cnt
        rec[i, 3] = sa
        rec[i, 4] = ma
        rec[i, 5] = cnt  # non-null count of max()'s input (no NULLs here)
        rec[i, 6] = np.float64(sv).view(np.int64)
    return rec


def _merge(records):
    """numpy restatement of k_table_merge for [add, add, max, add, add_f64]"""
    out = {}
    for rec in records:
        for r in rec:
            k = int(r[1])
            if k not in out:
                out[k] = [0, 0, 0, 0, 0.0]
            o = out[k]
            o[0] += int(r[2])
            o[1] += int(r[3])
            o[2] = max(o[2], int(r[4]))
            o[3] += int(r[5])
            o[4] += float(np.int64(r[6]).view(np.float64))
    return out


def _worker(rank, world, port, n_parts, rows, key, mode, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from eventql_amd import distributed as D
        import oracle_lib as O
        mine = D.partitions_for_rank(n_parts, rank, world)
        recs = []
        for p in mine:
            img = _write([_partition_columns(p, rows)])
            recs.append(_records_from_oracle(O.oracle_run(img, _plan(key))))
        # local pre-merge of this rank's partitions (one table per GPU)
        local = _merge(recs)
        dense = np.zeros((len(local), RW), dtype=np.int64)
        for i, (k, o) in enumerate(sorted(local.items())):
            dense[i] = [0, k, o[0], o[1], o[2], o[3], np.float64(o[4]).view(np.int64)]
        t = torch.from_numpy(dense.reshape(-1).copy())
        if mode == "all_gather":
            parts = D.exchange_all_gather(t, len(local), RW, max_groups=8192)
            # same exchange through persistent buffers with the records laid out
            # behind the count word (the RCCL path of bench.py)
            xbuf = D.exchange_buffers(RW, 8192, world, "cpu")
            xbuf[0][1:1 + t.numel()] = t
            parts2 = D.exchange_all_gather(None, len(local), RW, 8192, buffers=xbuf)
            assert [c for _, c in parts2] == [c for _, c in parts]
            assert all(torch.equal(a, b) for (a, _), (b, _) in zip(parts, parts2))
            # own records + everybody else's as one contiguous block (what bench.py
            # hands to evql_query_import_groups)
            foreign, cnt = D.gather_foreign(parts, rank, RW)
            assert cnt == sum(c for r, (_, c) in enumerate(parts) if r != rank)
            merged = _merge([dense] + ([foreign.numpy().reshape(cnt, RW)] if cnt else []))
            q.put((rank, merged))
        else:
            recv, cnt = D.exchange_all_to_all(t, len(local), RW)
            owned = _merge([recv.numpy().reshape(cnt, RW)])
            assert all(k % world == rank for k in owned), "received a key this rank does not own"
            q.put((rank, owned))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "ERR " + traceback.format_exc()))
        raise e


@pytest.mark.parametrize("mode,key", [("all_gather", "k"), ("all_to_all", "u")])
def test_two_rank_partial_aggregate_exchange(built, mode, key):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    world, n_parts, rows = 2, 3, 40_000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_parts, rows, key, mode, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        r, m = q.get(timeout=300)
        assert not isinstance(m, str), m
        results[r] = m
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # oracle on the concatenation of all partitions
    img = _write([_partition_columns(p, rows) for p in range(n_parts)])
    exp = {int(r[0]): r for r in O.oracle_run(img, _plan(key)).rows()}
    if mode == "all_gather":
        merged_views = [results[0], results[1]]  # every rank holds the full result
    else:
        union = {}
        for r in range(world):
            assert not (set(union) & set(results[r])), "key ranges overlap"
            union.update(results[r])
        merged_views = [union]
    for merged in merged_views:
        assert set(merged) == set(exp)
        for k, (_, cnt, sa, ma, sv) in exp.items():
            o = merged[k]
            assert (o[0], o[1], o[2]) == (cnt, sa, ma), (k, o, exp[k])
            assert abs(o[4] - sv) <= 1e-9 * abs(sv)


def test_partition_assignment():
    sys.path.insert(0, ROOT)
    from eventql_amd import distributed as D
    for n, w in ((8, 8), (8, 2), (3, 2), (1, 4), (10, 4)):
        seen = []
        for r in range(w):
            seen += D.partitions_for_rank(n, r, w)
        assert seen == list(range(n))
