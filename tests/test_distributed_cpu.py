"""world_size-2 `gloo` test (CPU) of the launcher-side plumbing of the N > 1 path:
partition assignment and the transport callbacks eventql_amd.distributed hands to
`evql_query_exchange` (csrc/exchange.cc) when no RCCL is available.  The callbacks get
raw addresses and word counts, exactly as the C side calls them; here the buffers are
host memory, on the GPU box device memory (tests/test_gpu_distributed.py).  The exchange
kernels themselves (bucketing, first-row resolution, rank-ordered merges) need a GPU:
tests/test_gpu_exchange.py, test_gpu_distributed.py."""
import ctypes as C
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


def _payload(src, dst, n):
    """the words rank `src` sends to rank `dst`"""
    return (np.arange(n, dtype=np.int64) * 1000003 + src * 7919 + dst * 104729) ^ (src << 40)


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from eventql_amd import distributed as D
        tr = D.GlooTransport(None, "cpu")
        # all_gather of the per-destination counts (two words per destination, as the
        # exchange sends them: records, string bytes), incl. values beyond 2^31
        mine = [rank * 10 + d for d in range(2 * world)] + [(1 << 40) + rank]
        got = tr.all_gather(mine)
        exp = []
        for r in range(world):
            exp += [r * 10 + d for d in range(2 * world)] + [(1 << 40) + r]
        assert got == exp, (got, exp)
        # variable all-to-all of words, one empty message, three rounds reusing buffers
        for rnd in range(3):
            counts = [[(5 + 3 * s + 11 * d + rnd) if (s, d) != (1, 0) else 0 for d in range(world)]
                      for s in range(world)]
            send = np.concatenate([_payload(rank, d, counts[rank][d]) + rnd for d in range(world)])
            recv_counts = [counts[s][rank] for s in range(world)]
            recv = np.full(sum(recv_counts) + 4, -1, dtype=np.int64)  # guard words behind
            tr.all_to_all(send.ctypes.data, counts[rank], recv.ctypes.data, recv_counts, None)
            off = 0
            for s in range(world):
                n = recv_counts[s]
                assert (recv[off:off + n] == _payload(s, rank, n) + rnd).all(), (rank, s, rnd)
                off += n
            assert (recv[off:] == -1).all()
        q.put((rank, "ok"))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "ERR " + traceback.format_exc()))
        raise e


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_transport_callbacks(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for _ in range(world):
        r, m = q.get(timeout=300)
        assert m == "ok", m
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0


def test_partition_assignment():
    sys.path.insert(0, ROOT)
    from eventql_amd import distributed as D
    for n, w in ((8, 8), (8, 2), (3, 2), (1, 4), (10, 4)):
        seen = []
        for r in range(w):
            seen += D.partitions_for_rank(n, r, w)
        assert seen == list(range(n))


def test_exchange_entry_points_are_exported(built):
    """the C ABI of the exchange step loads and exports every symbol (no compute call
    without a GPU)"""
    import eventql_amd as E
    L = E.lib()
    for name in ("evql_exchange_create", "evql_exchange_create_rccl", "evql_rccl_unique_id",
                 "evql_hub_create", "evql_hub_destroy", "evql_exchange_create_hub",
                 "evql_exchange_destroy", "evql_exchange_backend", "evql_query_exchange",
                 "evql_exchange_last_stats"):
        assert hasattr(L, name), name
    hub = E.Hub(2)
    hub.close()
