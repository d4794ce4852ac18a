#!/usr/bin/env python3
"""Scratch: one ORDER BY query through the probe, MODE cpu and MODE gpu, rows side by side."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refcases
PROBE = os.path.join(ROOT, "oracle", "_ref", "csql_probe")
qs = sys.argv[1:] or [
 "select k, count_distinct(k10) as x0, sum(k) as x1 from t where a >= 0 and b >= 0 and k >= 0 and p >= 0 and k10 >= 0 and a > 60000 group by k order by x0 desc, k limit 5;",
 "select k, count_distinct(k10) as x0, sum(k) as x1 from t where a >= 0 and b >= 0 and k >= 0 and p >= 0 and k10 >= 0 and a > 60000 group by k order by x0 desc, k limit 2000;",
 "select k, count_distinct(k10) as x0 from t where a > 60000 and k10 >= 0 and k >= 0 group by k order by x0, k limit 5;",
 "select k, count_distinct(a) as x0 from t where a > 60000 and k >= 0 group by k order by x0 desc, k limit 5;"]
img, _, kind = refcases.table_image("mixed")
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "t.cst"); open(path, "wb").write(img)
    out = {}
    for mode in ("cpu", "gpu"):
        cmds = ["TABLE t %s %s" % (path, kind), "ROWS on", "MODE " + mode] + ["SQL " + q for q in qs]
        p = subprocess.run([PROBE], input="\n".join(cmds) + "\n", capture_output=True, text=True)
        out[mode] = [json.loads(l) for l in p.stdout.splitlines() if l.strip()]
for q, c, g in zip(qs, out["cpu"], out["gpu"]):
    print(q); print("  cpu", c["rows"][:8]); print("  gpu", g["rows"][:8], g.get("decisions")); print("  equal", c["rows"] == g["rows"])
