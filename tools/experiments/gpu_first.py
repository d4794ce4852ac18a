import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as G
import eventql_amd as E
from eventql_amd import capi as K, synth, bench_plans as B
import oracle_lib as O

t0 = time.time(); G.smoke(); print("smoke", time.time() - t0, flush=True)
ctx = E.Context(0)
# device generator vs host twin
n = 1_000_000
t = ctx.generate(n, "kabvu", u_mod=10_000_000)
img = t.download_image()
c = synth.table_columns(n)
r = O.TableReader.__new__(O.TableReader)
open("/tmp/gen.cst", "wb").write(img)
rd = O.TableReader("/tmp/gen.cst", "orc")
for name in "kab":
    _, _, pr, v = rd.read(name, n, "uint"); assert (v == c[name]).all(), name
_, _, pr, v = rd.read("v", n, "float"); assert (v == c["v"]).all()
_, _, pr, v = rd.read("u", n, "uint"); assert (v == c["x"] % np.uint64(10_000_000)).all()
print("device generator matches host twin", flush=True)
for plan_fn in (B.config2, B.config3):
    p = plan_fn()
    got = t.query(p).run()
    exp = O.oracle_run(img, p)
    g = {r[0]: r for r in got.rows()}; e = {r[0]: r for r in exp.rows()}
    assert set(g) == set(e)
    for k in e:
        for a, b in zip(g[k], e[k]):
            if isinstance(b, float): assert abs(a - b) <= 1e-6 * abs(b), (k, g[k], e[k])
            else: assert a == b, (k, g[k], e[k])
    print(plan_fn.__name__, "parity OK", len(e), "groups", flush=True)
t.close()
for n in (100_000_000, 1_000_000_000):
    t0 = time.time(); t = ctx.generate(n, "kabv"); ctx.synchronize(); print("generate", n, time.time() - t0, flush=True)
    for plan_fn in (B.config2, B.config3):
        q = t.query(plan_fn())
        for it in range(4):
            q.execute(); s = q.stats()
            print(plan_fn.__name__, n, "kernel_ms %.3f" % s["kernel_ms"], "rows/s %.3e" % (n / s["kernel_ms"] * 1e3),
                  "GB/s %.1f" % (s["algorithmic_bytes"] / s["kernel_ms"] / 1e6), "groups", s["num_groups"], "passed", s["rows_passed"], flush=True)
    t.close()
