"""kernel time of config 2's query (k, sum(v), count(1) GROUP BY k) for very few groups:
the replicated LDS table (KernelPlan::lds_replicas) against 1000 groups.
usage: python tools/experiments/tiny_groups.py [rows]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eventql_amd as E
from eventql_amd import bench_plans as B

rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
ctx = E.Context(0)
for k_mod in (1, 2, 3, 4, 9, 16, 33, 64, 1000):
    t = ctx.generate(rows, "kv", k_mod=k_mod)
    for hint in (k_mod,):
        q = t.query(B.config2(groups_hint=hint))
        for _ in range(3):
            q.launch()
            q.finish()
        ms = []
        for _ in range(5):
            q.launch()
            q.finish()
            ms.append(q.stats()["kernel_ms"])
        st = q.stats()
        rep = "EVQL_LDS_REP 16" in q.kernel_source()
        print("groups %5d hint %5d: kernel %.3f ms  (%.0f GB/s)  replicated=%s est=%d" % (
            k_mod, hint, min(ms), 16.0 * rows / (min(ms) * 1e-3) / 1e9, rep, st["estimated_groups"]),
            flush=True)
        q.close()
    t.close()
