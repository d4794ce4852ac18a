import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import eventql_amd as E
from eventql_amd import capi as K, bench_plans as B
from eventql_amd.plan import *
ctx = E.Context(0)
n = int(os.environ.get("N", 500_000_000))
t = ctx.generate(n, "kabv")
S = B.SCHEMA
k, a, b, v = col("k"), col("a"), col("b"), col("v")
plans = {
 "c2 sum(v),count": dict(select=[k, sum_(v), count(1)], group_by=[k]),
 "count only": dict(select=[k, count(1)], group_by=[k]),
 "sum(v) only": dict(select=[k, sum_(v)], group_by=[k]),
 "sum(a) only (u64, 16B/row)": dict(select=[k, sum_(a)], group_by=[k]),
 "sum(a),count (u64)": dict(select=[k, sum_(a), count(1)], group_by=[k]),
 "global count": dict(select=[count(1)]),
 "global sum(v),count": dict(select=[sum_(v), count(1)]),
 "c3": dict(select=[k, sum_(v), count(1), sum_(b)], group_by=[k], where=(a > 30000) & (b < 30000)),
}
def run(name, kw, env):
    for kk in ("EVQL_FORCE_BLOCK","EVQL_FORCE_UNROLL","EVQL_FORCE_LDS_SLOTS"): os.environ.pop(kk, None)
    os.environ.update(env)
    q = t.query(Plan(S, groups_hint=1000, **kw))
    best = 1e9
    for i in range(4):
        q.execute(); best = min(best, q.stats()["kernel_ms"])
    s = q.stats()
    print("%-28s %-44s %.3f ms  %.1f GB/s  %.2e rows/s" % (name, env, best, s["algorithmic_bytes"]/best/1e6, n/best*1e3), flush=True)
    q.close()
for name, kw in plans.items():
    run(name, kw, {})
for env in ({"EVQL_FORCE_LDS_SLOTS":"2048"}, {"EVQL_FORCE_LDS_SLOTS":"1024"}, {"EVQL_FORCE_UNROLL":"2"}, {"EVQL_FORCE_UNROLL":"8"}, {"EVQL_FORCE_BLOCK":"512"}):
    run("c2 sum(v),count", plans["c2 sum(v),count"], env)
    run("c3", plans["c3"], env)
