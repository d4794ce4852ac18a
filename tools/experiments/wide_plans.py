import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import eventql_amd as E
from eventql_amd import bench_plans as B
from eventql_amd.plan import Plan, col, count, sum_, min_, max_, mean, If
ctx = E.Context(0)
t = ctx.generate(200_000_000, "kabv")
k, a, b, v = col("k"), col("a"), col("b"), col("v")
plans = {
 "wide": dict(select=[k, sum_(a), sum_(b), sum_(a * b), min_(a), max_(b), mean(v), sum_(v), count(1),
                      sum_(If(a > b, a - b, b - a)), max_(v * 2.0), min_(v)], group_by=[k], where=(a > 100) & (b < 65000)),
 "wider": dict(select=[k, sum_(a), sum_(b), sum_(a * b), min_(a), max_(b), mean(v), sum_(v), count(1),
                       sum_(If(a > b, a - b, b - a)), max_(v * 2.0), min_(v), mean(a), mean(b), sum_(a % 7), sum_(b % 13)],
               group_by=[k], where=(a > 100) & (b < 65000)),
}
for name, kw in plans.items():
    q = t.query(Plan(B.SCHEMA, groups_hint=1000, **kw))
    for _ in range(2):
        q.launch(); q.finish()
    ms = []
    for _ in range(5):
        q.launch(); q.finish(); ms.append(q.stats()["kernel_ms"])
    import re
    src = q.kernel_source()
    print(name, "unroll", re.search(r"#define EVQL_UNROLL (\d+)", src).group(1), "kernel_ms %.3f" % (sum(ms) / len(ms)))
    q.close()
