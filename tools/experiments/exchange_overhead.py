#!/usr/bin/env python3
"""Fixed cost of the exchange step on one GPU: a single-rank RCCL communicator (everything
is sent to itself), config 3 (1000 groups, GATHER_ALL) and config 4 (1e7 groups, BY_OWNER)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import eventql_amd as E  # noqa: E402
from eventql_amd import bench_plans as B, capi as K  # noqa: E402

ctx = E.Context(0)
x = E.Exchange.rccl(ctx, 1, 0, E.Exchange.rccl_unique_id())
for name, rows, plan, mode, gen in (
        ("config3", 200_000_000, B.config3(), K.EXCHANGE_GATHER_ALL, {}),
        ("config4", 125_000_000, B.config4(groups_hint=10_000_000), K.EXCHANGE_BY_OWNER,
         dict(u_mod=10_000_000))):
    t = ctx.generate(rows, "kabv" if name == "config3" else "uav", **gen)
    q = t.query(plan)
    for _ in range(2):
        q.launch(); q.finish(); q.exchange(x, mode)
    ctx.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        q.launch(); q.finish()
    ctx.synchronize()
    t1 = time.perf_counter()
    for _ in range(n):
        q.launch(); q.finish(); q.exchange(x, mode)
    ctx.synchronize()
    t2 = time.perf_counter()
    print(name, "scan %.3f ms  scan+exchange %.3f ms  exchange %.3f ms" %
          ((t1 - t0) / n * 1e3, (t2 - t1) / n * 1e3, ((t2 - t1) - (t1 - t0)) / n * 1e3), x.stats())
    q.close(); t.close()
x.close()
