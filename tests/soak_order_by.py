#!/usr/bin/env python3
"""Soak on an MI355X box: random `GROUP BY .. ORDER BY .. LIMIT .. OFFSET` queries through the
reference's engine, CPU operators vs the GPU operator with ORDER BY / LIMIT pushed into it
(evql_query_set_order: device top-k over the group records).  The last sort key is the
(unique) group key, so the order is fully specified and the rows must agree one by one.
usage: tests/soak_order_by.py <first seed> <count>"""
import json
import os
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refcases  # noqa: E402

PROBE = os.path.join(ROOT, "oracle", "_ref", "csql_probe")
KEYS = ["k", "b", "p", "k10", "a", "s", "w"]  # non-nullable columns of the mixed table (s: STRING)
AGGS = ["count(1)", "sum(a)", "sum(b)", "sum(k)", "sum(p)", "count_distinct(k10)", "sum(a * 3 + b)"]


def query(seed):
    r = random.Random(seed)
    key = r.choice(KEYS)
    aggs = r.sample(AGGS, r.randint(1, 3))
    sel = [key] + ["%s as x%d" % (a, i) for i, a in enumerate(aggs)]
    order = []
    for i in r.sample(range(len(aggs)), r.randint(0, min(2, len(aggs)))):
        order.append("x%d%s" % (i, r.choice(["", " desc", " asc"])))
    order.append(key + r.choice(["", " desc"]))
    where = " and ".join("%s >= 0" % c for c in ("a", "b", "k", "p", "k10", "w")) + " and (s = '' or s != '')"
    if r.random() < 0.5:
        where += " and a > %d" % r.choice([1000, 30000, 60000])
    sql = "select %s from t where %s group by %s order by %s" % (", ".join(sel), where, key, ", ".join(order))
    if r.random() < 0.8:
        sql += " limit %d" % r.choice([1, 5, 10, 100, 1000, 5000])
        if r.random() < 0.4:
            sql += " offset %d" % r.choice([1, 3, 50, 900])
    return sql + ";"


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    sqls = [query(s) for s in range(first, first + count)]
    img, _, kind = refcases.table_image("mixed")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "t.cst")
        open(path, "wb").write(img)
        for mode in ("cpu", "gpu"):
            cmds = ["TABLE t %s %s" % (path, kind), "ROWS on", "MODE " + mode] + ["SQL " + q for q in sqls]
            p = subprocess.run([PROBE], input="\n".join(cmds) + "\n", capture_output=True, text=True)
            if p.returncode != 0:
                raise SystemExit("probe failed in MODE %s: %s" % (mode, p.stderr[-2000:]))
            out[mode] = [json.loads(l) for l in p.stdout.splitlines() if l.strip()]
    bad = []
    pushed = 0
    for q, c, g in zip(sqls, out["cpu"], out["gpu"]):
        if not c["ok"] or not g["ok"]:
            if c["ok"] != g["ok"]:
                bad.append((q, c.get("error"), g.get("error")))
            continue
        d = {x["node"]: x["lowered"] for x in g.get("decisions", [])}
        pushed += 1 if d.get("orderby") else 0
        if c["types"] != g["types"] or c["rows"] != g["rows"]:
            bad.append((q, len(c["rows"]), len(g["rows"])))
    print(json.dumps(dict(queries=len(sqls), order_by_on_device=pushed, mismatches=len(bad))))
    for b in bad[:10]:
        print("MISMATCH", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
