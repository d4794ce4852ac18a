import os
import sys
sys.path.insert(0, "tests")
import numpy as np
import eventql_amd as E
from eventql_amd import capi as K
from eventql_amd.plan import Plan, col, count, sum_, max_, min_
import oracle_lib as O

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 300_001
rng = np.random.default_rng(bits)
maxv = (1 << bits) - 1
x = rng.integers(0, maxv + 1, n, dtype=np.uint64)
y = rng.integers(0, maxv + 1, n, dtype=np.uint64)


def table(enc):
    w = E.Writer([dict(name="x", logical_type=K.COL_UNSIGNED_INT, storage_type=enc,
                       bitpack_max_value=maxv),
                  dict(name="y", logical_type=K.COL_UNSIGNED_INT, storage_type=enc,
                       bitpack_max_value=maxv)])
    w.put("x", x)
    w.put("y", y)
    w.commit(n)
    return w.image()


ctx = E.Context(0)
S = dict(x=K.T_UINT64, y=K.T_UINT64)
W = col("y") >= (maxv // 3)
G = col("x") % 13
variants = [
    ("full", dict(select=[G, count(1), sum_(col("y")), max_(col("x"))], group_by=[G], where=W)),
    ("nomax", dict(select=[G, count(1), sum_(col("y"))], group_by=[G], where=W)),
    ("onlymax", dict(select=[G, max_(col("x"))], group_by=[G], where=W)),
    ("nowhere", dict(select=[G, count(1), sum_(col("y")), max_(col("x"))], group_by=[G])),
    ("minx", dict(select=[G, count(1), min_(col("x"))], group_by=[G], where=W)),
]
for enc_name, enc in (("bitpacked", K.ENC_UINT32_BITPACKED), ("plain64", K.ENC_UINT64_PLAIN)):
    img = table(enc)
    t = ctx.open_image(img)
    for env in ({}, {"EVQL_FORCE_LDS_SLOTS": "0"}):
        for k2, v2 in env.items():
            os.environ[k2] = v2
        for name, kw in variants:
            plan = Plan(S, **kw)
            exp = sorted(O.oracle_run(img, plan).rows(), key=repr)
            q = t.query(plan)
            got = sorted(q.run().rows(), key=repr)
            ok = got == exp
            print(enc_name, env, name, "OK" if ok else "MISMATCH", q.stats()["num_groups"])
            if not ok:
                print("   extra:", [r for r in got if r not in exp][:4])
            q.close()
        for k2 in env:
            del os.environ[k2]
    t.close()
