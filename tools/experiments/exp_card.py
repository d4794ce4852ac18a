"""group-cardinality sweep of the config-2 / config-3 shapes (k_mod groups)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import eventql_amd as E
from eventql_amd import bench_plans as B
from eventql_amd.plan import *
ctx = E.Context(0)
n = int(os.environ.get("N", 400_000_000))
S = B.SCHEMA
k, a, b, v = col("k"), col("a"), col("b"), col("v")
for kmod in (1, 2, 4, 16, 64, 256, 1000, 4000, 20000, 100000, 1000000):
    t = ctx.generate(n, "kabv", k_mod=kmod)
    for name, kw in (("c2", dict(select=[k, sum_(v), count(1)], group_by=[k])),
                     ("c3", dict(select=[k, sum_(v), count(1), sum_(b)], group_by=[k], where=(a > 30000) & (b < 30000)))):
        for hint in (kmod, 0):
            q = t.query(Plan(S, groups_hint=hint, **kw))
            best = 1e9
            for i in range(3):
                q.execute(); best = min(best, q.stats()["kernel_ms"])
            s = q.stats()
            print("%s groups=%-8d hint=%-8d lds=%d  %.3f ms  %.1f GB/s  %.2e rows/s" % (name, kmod, hint, s["used_lds_table"], best, s["algorithmic_bytes"]/best/1e6, n/best*1e3), flush=True)
            q.close()
    t.close()
